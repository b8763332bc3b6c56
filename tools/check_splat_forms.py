#!/usr/bin/env python3
"""The binned and the one-pass form of trt_splat_dev give the same image bit for bit — on images at the limits of the
binned form (8,192 bins, 16,383 pixels a side) and beyond them (where both calls take the one-pass form)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402
import torch
from toroidal_ray_tracing_amd import camera
from toroidal_ray_tracing_amd.tracer import Tracer
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
for W, H in ((8192, 8192), (16383, 4000), (16384, 64), (5000, 3000)):
    n = 2_000_000
    gen = torch.Generator(device=dev).manual_seed(5)
    cloud = torch.zeros(n, 8, device=dev)
    cloud[:, :3] = torch.rand(n, 3, device=dev, generator=gen) * 6 - 3
    cloud[:, 4:7] = torch.rand(n, 3, device=dev, generator=gen)
    cloud[n // 2:n // 2 + 1000] = cloud[:1000]
    vp = camera.perspective_vk(60, W / H) @ camera.look_at((1.0, 2.0, 7.0), (0.0, 0.0, 0.0))
    a = torch.empty(H, W, 4, device=dev); b = torch.empty(H, W, 4, device=dev)
    os.environ.pop("TRT_SPLAT_VARIANT", None); _tuning.reload(tr)
    for ps in (2.5, 17.0):
        tr.splat_dev(cloud.data_ptr(), n, vp, W, H, a.data_ptr(), point_size=ps, stream=s.cuda_stream)
        os.environ["TRT_SPLAT_VARIANT"] = "0"; _tuning.reload(tr)
        tr.splat_dev(cloud.data_ptr(), n, vp, W, H, b.data_ptr(), point_size=ps, stream=s.cuda_stream)
        os.environ.pop("TRT_SPLAT_VARIANT"); _tuning.reload(tr)
        torch.cuda.synchronize()
        print(W, H, ps, "identical" if torch.equal(a.view(torch.int32), b.view(torch.int32)) else "DIFFERENT", "covered", float((a[..., 0] != 0.8).float().mean()))
    del a, b, cloud
