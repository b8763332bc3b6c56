#!/usr/bin/env python3
"""trace(rays_in -> hits_out) timing over grid sizes (TRT_TRACE_BLOCKS): BASELINE config 2 rays and random aimed rays."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402  (loads the -DTRT_TUNING build, see _tuning.py)
import torch
from toroidal_ray_tracing_amd import abi, camera
from toroidal_ray_tracing_amd.tracer import Tracer
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
W = H = 2048
sc, g, pc = camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(1)
rend = torch.empty(W * H, 16, device=dev)
tr.render_dev(sc, g, pc, W, H, 0, rendered_ptr=rend.data_ptr(), stream=s.cuda_stream)
r = rend.view(W, H, 16).permute(1, 0, 2).reshape(-1, 16)
rays_a = [r[:, 8 + k].contiguous() for k in range(3)] + [r[:, 12 + k].contiguous() for k in range(3)]
n = W * H
gen = torch.Generator(device=dev).manual_seed(1)
o = torch.rand(n, 3, device=dev, generator=gen) * 8 - 4
tgt = torch.randn(n, 3, device=dev, generator=gen); tgt = tgt / tgt.norm(dim=1, keepdim=True) * (torch.rand(n, 1, device=dev, generator=gen) * 1.2)
d = tgt - o; d = d / d.norm(dim=1, keepdim=True)
rays_b = [o[:, k].contiguous() for k in range(3)] + [d[:, k].contiguous() for k in range(3)]
out = {k: torch.empty(n, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
op = {k: v.data_ptr() for k, v in out.items()}
def t(rays, reps=20, rounds=5):
    rp = [a.data_ptr() for a in rays]; res = []
    for k in range(rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): tr.trace_dev(sc, rp, n, op, stream=s.cuda_stream)
        e1.record(s); torch.cuda.synchronize()
        if k: res.append(e0.elapsed_time(e1) / reps)
    return statistics.median(res)
for b in (1024, 2048, 4096, 8192, 16384):
    os.environ["TRT_TRACE_BLOCKS"] = str(b); _tuning.reload(tr)
    a, c = t(rays_a), t(rays_b)
    print(f"blocks {b:6d}: camera rays {a:.4f} ms ({52*n/a/1e6:.0f} GB/s)   aimed rays {c:.4f} ms ({52*n/c/1e6:.0f} GB/s)")
