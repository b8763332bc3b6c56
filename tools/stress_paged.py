#!/usr/bin/env python3
"""Paged scatter of trt_splat_dev under the clouds that stress its page bookkeeping — everything in one bin, big points on bin
corners (four records each), points in screen order (a block's run spans many pages), tiny and large clouds alternating through
ONE context — against the one-pass form (TRT_SPLAT_VARIANT=0) bit for bit.  usage: stress_paged.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402
import torch
from toroidal_ray_tracing_amd import camera
from toroidal_ray_tracing_amd.tracer import Tracer
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
slow = len(sys.argv) > 2 and sys.argv[2] == "slow"   # TRT_DEBUG_SKIP=128: waiters look again only after ~50 us, the window moves past them
gen = torch.Generator(device=dev).manual_seed(11)
def cloud_of(n, kind, W, H):
    c = torch.zeros(n, 8, device=dev)
    u = torch.rand(n, 3, device=dev, generator=gen)
    if kind == "cluster":      # a blob that lands inside one bin
        c[:, :3] = (u - 0.5) * 0.05 + torch.tensor([0.3, 0.2, 0.0], device=dev)
    elif kind == "corner":     # around the view axis: the centre of the image is a bin corner when W/128 and H/64 are even
        c[:, :3] = (u - 0.5) * 0.02
    elif kind == "sheet":      # a plane facing the camera, points in row-major order of the plane
        k = torch.arange(n, device=dev)
        side = int(n ** 0.5) + 1
        c[:, 0] = ((k % side).float() / side - 0.5) * 5
        c[:, 1] = ((k // side).float() / side - 0.5) * 5
        c[:, 2] = u[:, 2] * 0.01
    else:
        c[:, :3] = u * 6 - 3
    c[:, 4:7] = torch.rand(n, 3, device=dev, generator=gen)
    return c
bad = 0; cases = 0
for rnd in range(rounds):
    for W, H in ((2048, 2048), (256, 128), (64, 64), (4096, 2048), (1000, 700)):
        for kind in ("cluster", "corner", "sheet", "random"):
            for n, ps in ((1000, 2.5), (70_000, 31.0), (300_000, 2.5), (2_000_000, 1.0), (2_000_000, 9.0), (5, 2.5)):
                if W * H > 4_000_000 and n < 70_000: continue
                c = cloud_of(n, kind, W, H)
                vp = camera.perspective_vk(60, W / H) @ camera.look_at((0.0, 0.0, 5.0), (0.0, 0.0, 0.0))
                a = torch.empty(H, W, 4, device=dev); b = torch.empty(H, W, 4, device=dev)
                os.environ.pop("TRT_SPLAT_VARIANT", None)
                if slow: os.environ["TRT_DEBUG_SKIP"] = "128"
                _tuning.reload(tr)
                tr.splat_dev(c.data_ptr(), n, vp, W, H, a.data_ptr(), point_size=ps, stream=s.cuda_stream)
                os.environ.pop("TRT_DEBUG_SKIP", None)
                os.environ["TRT_SPLAT_VARIANT"] = "0"; _tuning.reload(tr)
                tr.splat_dev(c.data_ptr(), n, vp, W, H, b.data_ptr(), point_size=ps, stream=s.cuda_stream)
                torch.cuda.synchronize()
                same = torch.equal(a.view(torch.int32), b.view(torch.int32)); cases += 1
                if not same:
                    bad += 1
                    print(f"DIFFERENT: {W}x{H} {kind} n={n} point_size={ps}: {(a != b).any(dim=2).sum().item()} pixels", flush=True)
                del a, b, c
    print(f"round {rnd}: {cases} cases, {bad} different", flush=True)
print(f"stress_paged: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
