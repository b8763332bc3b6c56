#!/usr/bin/env python3
"""Long differential run of trt_splat_dev against the oracle's sequential rasteriser (GPU box): N random clouds (size,
extent, duplicates = depth ties, NaN / lowest() points), random cameras, image sizes (ragged bins), point sizes on both
sides of the binned form's limit — the image bit for bit.  One context: the bin counters and scratch are reused.
usage: fuzz_splat.py [N=300] [seed=1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import oracle   # the checker, never the thing measured
from toroidal_ray_tracing_amd import camera
from toroidal_ray_tracing_amd.tracer import Tracer

n_rounds, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 300), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
oracle.lib()
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(seed)
bad, t0 = 0, time.time()
for k in range(n_rounds):
    W, H = int(rng.integers(8, 700)), int(rng.integers(8, 500))
    if k % 10 == 9:   # the other scatter kernels: up to 512 bins / up to 2,048 bins (both LDS-sorted) / more (direct)
        W, H = [(2500, 1400), (4096, 2048), (3000, 4000), (8192, 4100)][(k // 10) % 4]
    n = int(rng.choice([0, 1, 63, 1000, 40_000, 250_000]))
    pts = np.zeros((n, 8), np.float32)
    ext = float(rng.uniform(0.5, 6.0))
    pts[:, :3] = rng.uniform(-ext, ext, (n, 3))
    pts[:, 4:7] = rng.uniform(0, 1, (n, 3))
    if n >= 1000:
        m = n // 10
        pts[n // 2:n // 2 + m] = pts[:m]
        pts[n // 2:n // 2 + m, 4:7] = rng.uniform(0, 1, (m, 3))      # equal depth, different colour: the earlier point wins
        pts[::97, :3] = np.finfo(np.float32).min
        pts[5::101, int(rng.integers(0, 3))] = np.nan
    eye = rng.normal(size=3); eye = eye / np.linalg.norm(eye) * rng.uniform(0.5, 9.0)
    vp = camera.perspective_vk(float(rng.uniform(20, 120)), W / H) @ camera.look_at(tuple(eye), tuple(rng.uniform(-1, 1, 3)))
    ps = float(rng.choice([1.0, 2.5, 2.5, 7.0, 31.9, 33.0, 40.0]))
    clear = tuple(rng.uniform(0, 1, 3)) + (1.0,)
    d_pts = torch.from_numpy(pts).to(dev) if n else torch.zeros(1, 8, device=dev)
    out = torch.full((H, W, 4), -5.0, device=dev)
    tr.splat_dev(d_pts.data_ptr(), n, vp, W, H, out.data_ptr(), clear=clear, point_size=ps, stream=s)
    torch.cuda.synchronize()
    want = oracle.splat(pts, vp, W, H, clear=clear, point_size=ps)
    if not np.array_equal(out.cpu().numpy().view(np.uint32), want.view(np.uint32)):
        bad += 1
        print(f"MISMATCH round {k}: {W}x{H}, {n} points, point size {ps}", flush=True)
    if k % 50 == 49:
        print(f"{k + 1} rounds, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
tr.close()
print(f"fuzz_splat: {n_rounds} rounds, {bad} mismatches")
sys.exit(1 if bad else 0)
