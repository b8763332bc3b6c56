#!/usr/bin/env python3
"""Re-projection (trt_splat_dev) timing over grid sizes (TRT_SPLAT_BLOCKS_PER_CU)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402  (loads the -DTRT_TUNING build, see _tuning.py)
import torch
from toroidal_ray_tracing_amd import camera
from toroidal_ray_tracing_amd.tracer import Tracer
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
n = 4096 * 2048
gen = torch.Generator(device=dev).manual_seed(2)
cloud = torch.zeros(n, 8, device=dev)
cloud[:, :3] = torch.rand(n, 3, device=dev, generator=gen) * 6 - 3
cloud[:, 4:7] = torch.rand(n, 3, device=dev, generator=gen)
vp = camera.perspective_vk(60, 1.0) @ camera.look_at((1.0, 2.0, 7.0), (0.0, 0.0, 0.0))
img = torch.empty(2048, 2048, 4, device=dev)
def t(reps=5, rounds=4):
    res = []
    for k in range(rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): tr.splat_dev(cloud.data_ptr(), n, vp, 2048, 2048, img.data_ptr(), stream=s.cuda_stream)
        e1.record(s); torch.cuda.synchronize()
        if k: res.append(e0.elapsed_time(e1) / reps)
    return statistics.median(res)
for b in (4, 16, 64, 256):
    os.environ["TRT_SPLAT_BLOCKS_PER_CU"] = str(b); _tuning.reload(tr)
    print(f"random points, blocks/CU {b:4d}: {t():.4f} ms")
os.environ.pop("TRT_SPLAT_BLOCKS_PER_CU"); _tuning.reload(tr)
# the real pipeline: a 4096x2048 toroidal capture (points in the capture's x*H+y order) re-projected
from toroidal_ray_tracing_amd import abi
W, H = 4096, 2048
sc = camera.single_torus_scene(R=6.0, r=1.5, material=camera.PLASTIC)
g, pc = camera.toroidal_camera(W, H), abi.make_push(max_depth=3, rho=4.0)
rend = torch.empty(W * H, 16, device=dev)
tr.render_dev(sc, g, pc, W, H, 0, camera=1, rendered_ptr=rend.data_ptr(), stream=s.cuda_stream)
cloud[:, :3] = rend[:, 0:3]; cloud[:, 3] = 0; cloud[:, 4:7] = rend[:, 4:7]; cloud[:, 7] = 0
vp = camera.perspective_vk(60, 1.0) @ camera.look_at((0.5, 1.0, -1.0), (6.0, 0.0, 2.0))
print(f"captured points (8.4 M, coherent order) -> 2048^2: {t():.4f} ms; covered {100 * (img[..., :3] != 0.8).any(dim=2).float().mean().item():.0f}% of the view")
