#!/usr/bin/env python3
"""MEASURED bound of what re-ordering rays can buy trt_trace on incoherent rays: bench.py's aimed set (2^20 rays) traced in its
own order and in orders that only an oracle could produce — sorted by each ray's TRUE walk length (polynomial evaluations of
the CPU restatement; tools/_aimed_ev.npz, made by tools/make_aimed_ev.py on the CPU) inside windows of W rays and over the
whole set.  Same rays, same arithmetic, permuted consistently: only the company a ray keeps in its wave changes.
usage: trace_sorted.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from toroidal_ray_tracing_amd import camera
from toroidal_ray_tracing_amd.tracer import Tracer
_ev = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_aimed_ev.npz")
if not os.path.exists(_ev):   # (not shipped to the GPU box: a few seconds of CPU)
    import subprocess
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "make_aimed_ev.py")], check=True)
z = np.load(_ev)
o, d, ev = z["o"], z["d"], z["ev"]
n = len(ev)
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
sc = camera.single_torus_scene()
out = {k: torch.empty(n, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
op = {k: v.data_ptr() for k, v in out.items()}
def timed(order):
    rays = [torch.from_numpy(np.ascontiguousarray(a[order, k])).to(dev) for a in (o, d) for k in range(3)]
    rp = [r.data_ptr() for r in rays]
    res = []
    for k in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): tr.trace_dev(sc, rp, n, op, stream=s.cuda_stream)
        e1.record(s); torch.cuda.synchronize()
        if k: res.append(e0.elapsed_time(e1) / 20)
    return statistics.median(res)
ident = np.arange(n)
base = timed(ident)
waves = ev.reshape(-1, 64)
print(f"{n} aimed rays: {100 * (ev == 0).mean():.1f} % culled in setup, {ev[ev > 0].mean():.2f} evaluations per solved test, slowest lane of a wave "
      f"{waves.max(1).mean():.2f}; as generated: {base:.4f} ms ({52 * n / base / 1e6:.0f} GB/s = {52 * n / base / 8e9 * 100:.1f} % of 8 TB/s)", flush=True)
for W in (256, 1024, 4096, 65536, n):
    order = np.concatenate([s0 + np.argsort(ev[s0:s0 + W], kind="stable") for s0 in range(0, n, W)])
    t = timed(order)
    print(f"sorted by the true walk length inside windows of {W:>7d} rays: {t:.4f} ms = {base / t:.3f}x ({52 * n / t / 8e9 * 100:.1f} % of 8 TB/s)", flush=True)
t = timed(ident)
print(f"as generated, again: {t:.4f} ms")
