#!/usr/bin/env python3
"""Re-projection: where resolve's time goes (TRT_DEBUG_SKIP bits 16 = no colour gather, 32 = no depth test, 64 = no record
loads; tuning build only, the image is then wrong).  usage: ablate_splat.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402
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
def t(reps=10, rounds=5):
    res = []
    for k in range(rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): tr.splat_dev(cloud.data_ptr(), n, vp, 2048, 2048, img.data_ptr(), stream=s.cuda_stream)
        e1.record(s); torch.cuda.synchronize()
        if k: res.append(e0.elapsed_time(e1) / reps)
    return statistics.median(res)
skips = [int(x) for x in sys.argv[1:]] or [0, 16, 32, 48, 64 + 32, 64 + 32 + 16]
for rnd in range(2):
    for skip in skips:
        os.environ["TRT_DEBUG_SKIP"] = str(skip); _tuning.reload(tr)
        print(f"skip {skip:3d}: {t():.4f} ms", flush=True)
