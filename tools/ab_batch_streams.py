#!/usr/bin/env python3
"""Interleaved A/B in ONE process: a 1/N part of the config-3 frame, N frames per launch, on one stream and with the launches
alternating over two (and three) streams, beside the full frame on one stream — rounds alternate between the candidates so
that the chip's state (clock, box) is the same for all.  usage: ab_batch_streams.py [--parts 8] [--rounds 12]"""
import argparse, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toroidal_ray_tracing_amd import abi, camera
from toroidal_ray_tracing_amd import distributed as trtd
from toroidal_ray_tracing_amd.tracer import Tracer
ap = argparse.ArgumentParser()
ap.add_argument("--parts", type=int, default=8)
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--frames", type=int, default=192)
a = ap.parse_args()
dev = torch.device("cuda:0"); W = H = 4096; n = a.parts
sc, g, pc = camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(5)
cur = torch.cuda.current_stream()
hits = ("t", "px", "py", "pz", "nx", "ny", "nz")
G = trtd.default_group_rows(H, n, trtd.DEFAULT_CYCLES)
cands = {"full frame, 1 stream": trtd.TiledFrame([Tracer(0)], W, H, 1, 0, dev, want_hits=hits, gather="none")}
for K in (1, 2, 3):
    cands[f"1/{n} part, {n} per launch, {K} stream(s)"] = trtd.TiledFrame([Tracer(0) for _ in range(K)], W, H, n, 0, dev, want_hits=hits,
                                                                         gather="none", group_rows=G, batch=n)
F = a.frames
def run(fr):
    for _ in range(F): fr.render(sc, g, pc, abi.TRT_CAMERA_PINHOLE, cur)
    fr.join(cur)
for fr in cands.values(): run(fr)
torch.cuda.synchronize()
out = {k: [] for k in cands}
for r in range(a.rounds):
    for k, fr in cands.items():
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(fr); torch.cuda.synchronize()
        out[k].append((time.perf_counter() - t0) / F * 1e6)
full = statistics.median(out["full frame, 1 stream"])
for k, v in out.items():
    m = statistics.median(v)
    rel = "" if k.startswith("full") else f"  = {full / n / m:.3f} of the full frame's {full:.1f} us / {n}"
    print(f"{k}: {m:.2f} us per frame (median of {a.rounds} interleaved rounds of {F} frames; min {min(v):.2f}, max {max(v):.2f}){rel}", flush=True)
