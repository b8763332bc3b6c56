#!/usr/bin/env python3
"""Do the eight output streams of a frame (rgba + t,P,N: 2^28 and 7 x 2^26 bytes at 4096^2) alias in the memory system when
they sit at power-of-two distances?  Times the config-3 frame with the first-hit streams carved out of one allocation at
base + k*(size + skew) for several skews.  usage: skew_streams.py [size]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402
import torch
from toroidal_ray_tracing_amd import camera
from toroidal_ray_tracing_amd.tracer import Tracer
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
sc, g, pc = camera.single_torus_scene(), camera.baseline_camera(W, W), camera.baseline_push(5)
names = ("t", "px", "py", "pz", "nx", "ny", "nz")
n = W * W
def timeit(fn, rounds=7, reps=20):
    for _ in range(5): fn()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): fn()
        e1.record(s); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return statistics.median(out), min(out)
for rep in range(2):
    for skew in (0, 256, 1024, 4096 + 256, 65536 + 256, 2 * 1024 * 1024 + 4096 + 256):
        pool = torch.empty(n * 4 * 4 + 7 * (n * 4 + skew) + 4096, dtype=torch.uint8, device=dev)
        base = pool.data_ptr()
        base += (-base) % 4096
        rgba_ptr = base
        hp = {k: base + n * 16 + skew + i * (n * 4 + skew) for i, k in enumerate(names)}
        ms = timeit(lambda: tr.render_dev(sc, g, pc, W, W, rgba_ptr, hit_ptrs=hp, stream=s.cuda_stream))
        print(f"skew {skew:8d} B  base%2MiB={base % (1<<21):7d}  {ms[0]:.4f} ms (min {ms[1]:.4f})  {44*n/ms[0]/1e6:6.0f} GB/s", flush=True)
        del pool
