#!/usr/bin/env python3
"""Does overlapping consecutive frames on K streams (one trt_ctx + one output set per stream) lift the latency floor of a
1/N part?  usage: bench_tiled_streams.py [--parts 8] [--part 0] [--streams 1 2 3 4] [--scene nested] [--f64]"""
import argparse, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toroidal_ray_tracing_amd import abi, camera
from toroidal_ray_tracing_amd import distributed as trtd
from toroidal_ray_tracing_amd.tracer import Tracer

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--parts", type=int, default=8)
ap.add_argument("--part", type=int, default=0)
ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 3, 4])
ap.add_argument("--scene", default="single", choices=["single", "nested"], help="nested: the eight nested tori of config 4")
ap.add_argument("--f64", action="store_true", help="FP64 solve (config 4)")
a = ap.parse_args()
dev = torch.device("cuda:0")
W = H = a.size
sc = camera.single_torus_scene() if a.scene == "single" else camera.nested_tori_scene()
g, pc = camera.baseline_camera(W, H), camera.baseline_push(5)
n = a.parts
G = trtd.default_group_rows(H, n, trtd.DEFAULT_CYCLES) if n > 1 else H
t = abi.trt_tiling(G, n, a.part, 1 if n > 1 else 0)
for K in a.streams:
    trs = [Tracer(0) for _ in range(K)]
    if a.f64:
        for tr_ in trs:
            tr_.set_solver(abi.TRT_SOLVE_F64)
    streams = [torch.cuda.Stream(device=dev) for _ in range(K)]
    rows = trs[0].tiling_rows(t, H) if n > 1 else H
    bufs = [(torch.empty(rows, W, 4, device=dev), {k: torch.empty(rows * W, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}) for _ in range(K)]
    ptrs = [(r.data_ptr(), {k: v.data_ptr() for k, v in h.items()}) for r, h in bufs]

    def frame(i):
        k = i % K
        if n > 1:
            trs[k].render_tiled_dev(sc, g, pc, W, H, t, ptrs[k][0], hit_ptrs=ptrs[k][1], stream=streams[k].cuda_stream)
        else:
            trs[k].render_dev(sc, g, pc, W, H, ptrs[k][0], hit_ptrs=ptrs[k][1], stream=streams[k].cuda_stream)
    for i in range(16):
        frame(i)
    torch.cuda.synchronize()
    out = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(256):
            frame(i)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 256)
    ms = statistics.median(out) * 1e3
    print(f"{a.scene}{' f64' if a.f64 else ''}: {n} parts, part {a.part}, {K} stream(s): {ms * 1e3:.1f} us per frame (wall, 256 frames)", flush=True)
    for tr in trs:
        tr.close()
