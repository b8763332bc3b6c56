#!/usr/bin/env python3
"""What one rank of an N-GPU job spends per frame on its 1/N part, with K frames in flight on K streams (one trt_ctx +
one output set per stream) — launched eagerly through Python + the C ABI, and replayed as ONE hipGraph per batch
(--graph: TiledFrame.capture_step, what bench.py --gpus N does), so that the host's share of a frame can be told from
the GPU's.  Wall clock per frame over batches of --frames frames; run it under `rocprofv3 --kernel-trace` and feed the
trace to tools/tiled_trace_report.py for the GPU-side figures (kernel durations, gaps, span per frame).
usage: bench_tiled_streams.py [--parts 8] [--part 0] [--streams 1 2 3 4] [--graph] [--both] [--scene nested] [--f64]"""
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
ap.add_argument("--streams", type=int, nargs="*", default=[1, 2, 3, 4])
ap.add_argument("--frames", type=int, default=192, help="frames per batch (a multiple of every --streams value)")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--batch", type=int, nargs="*", default=[], help="also: B frames per launch on ONE stream (trt_render_batch_dev)")
ap.add_argument("--batch-streams", type=int, nargs="*", default=[1], help="... and on K streams, the launches taking them in turn")
ap.add_argument("--graph", action="store_true", help="replay each batch as one hipGraph")
ap.add_argument("--both", action="store_true", help="eager and graph, interleaved rounds in one process")
ap.add_argument("--scene", default="single", choices=["single", "nested"], help="nested: the eight nested tori of config 4")
ap.add_argument("--f64", action="store_true", help="FP64 solve (config 4)")
a = ap.parse_args()
dev = torch.device("cuda:0")
W = H = a.size
sc = camera.single_torus_scene() if a.scene == "single" else camera.nested_tori_scene()
g, pc = camera.baseline_camera(W, H), camera.baseline_push(5)
n = a.parts
G = trtd.default_group_rows(H, n, trtd.DEFAULT_CYCLES) if n > 1 else None
cur = torch.cuda.current_stream()
for K, B in [(k, 1) for k in a.streams] + [(k, b) for b in a.batch for k in a.batch_streams]:
    F = a.frames - a.frames % (K * B)
    trs = [Tracer(0) for _ in range(K)]
    if a.f64:
        for tr_ in trs:
            tr_.set_solver(abi.TRT_SOLVE_F64)
    frame = trtd.TiledFrame(trs, W, H, n, a.part, dev, want_hits=("t", "px", "py", "pz", "nx", "ny", "nz"), gather="none", group_rows=G, batch=B)
    for _ in range(2 * K * B):
        frame.render(sc, g, pc, abi.TRT_CAMERA_PINHOLE, cur)
    frame.restart()
    modes = ["eager", "graph"] if a.both else (["graph"] if a.graph else ["eager"])
    if "graph" in modes:
        frame.capture_step(sc, g, pc, abi.TRT_CAMERA_PINHOLE, cur, F)

    def batch(mode):
        if mode == "graph":
            frame.step(cur)
        else:
            for _ in range(F):
                frame.render(sc, g, pc, abi.TRT_CAMERA_PINHOLE, cur)
            frame.join(cur)

    out = {m: [] for m in modes}
    for m in modes:
        batch(m)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for m in modes:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            batch(m)
            torch.cuda.synchronize()
            out[m].append((time.perf_counter() - t0) / F)
    for m in modes:
        us = statistics.median(out[m]) * 1e6
        print(f"{a.scene}{' f64' if a.f64 else ''}: {n} parts, part {a.part}, {K} stream(s), {B} frame(s) per launch, {m}: {us:.1f} us per frame "
              f"(wall, batches of {F} frames, median of {a.rounds}; min {min(out[m]) * 1e6:.1f})", flush=True)
    frame.finish()
    del frame
    for tr in trs:
        tr.close()
