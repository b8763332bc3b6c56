#!/usr/bin/env python3
"""Driver for profiling trt_trace_dev under rocprofv3: N launches on 2048² rays.
usage: run_trace.py [aimed|camera] [launches] [f64]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402
import torch
from toroidal_ray_tracing_amd import abi, camera
from toroidal_ray_tracing_amd.tracer import Tracer

kind = sys.argv[1] if len(sys.argv) > 1 else "aimed"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
if len(sys.argv) > 3 and sys.argv[3] == "f64":
    tr.set_solver(abi.TRT_SOLVE_F64)
W = H = 2048
n = W * H
sc = camera.single_torus_scene() if os.environ.get("SCENE", "single") == "single" else camera.nested_tori_scene()
if kind == "camera":
    g, pc = camera.baseline_camera(W, H), camera.baseline_push(1)
    rend = torch.empty(n, 16, device=dev)
    tr.render_dev(sc, g, pc, W, H, 0, rendered_ptr=rend.data_ptr(), stream=s.cuda_stream)
    r = rend.view(W, H, 16).permute(1, 0, 2).reshape(-1, 16)
    rays = [r[:, 8 + k].contiguous() for k in range(3)] + [r[:, 12 + k].contiguous() for k in range(3)]
else:
    gen = torch.Generator(device=dev).manual_seed(1)
    o = torch.rand(n, 3, device=dev, generator=gen) * 8 - 4
    tgt = torch.randn(n, 3, device=dev, generator=gen)
    tgt = tgt / tgt.norm(dim=1, keepdim=True) * (torch.rand(n, 1, device=dev, generator=gen) * 1.2)
    d = tgt - o
    d = d / d.norm(dim=1, keepdim=True)
    rays = [o[:, k].contiguous() for k in range(3)] + [d[:, k].contiguous() for k in range(3)]
out = {k: torch.empty(n, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
op = {k: v.data_ptr() for k, v in out.items()}
rp = [a.data_ptr() for a in rays]
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    tr.trace_dev(sc, rp, n, op, stream=s.cuda_stream)
e0.record(s)
for _ in range(reps):
    tr.trace_dev(sc, rp, n, op, stream=s.cuda_stream)
e1.record(s)
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"{kind} rays: {ms:.4f} ms/launch  {52 * n / ms / 1e6:.0f} GB/s ({52 * n / ms / 8e9 * 100:.1f} % of 8 TB/s)  hit fraction {torch.isfinite(out['t']).float().mean().item():.3f}")
