#!/usr/bin/env python3
"""A 1/N part of the config-3 frame with K frames in flight (tools/bench_tiled_streams.py) against launch-shape knobs of
the -DTRT_TUNING build, interleaved rounds in ONE process.  usage: part_knobs.py [--parts 8] [--streams 4] KNOB=V[,KNOB=V] …
(an empty spec "" = the defaults)"""
import argparse, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402
import torch
from toroidal_ray_tracing_amd import abi, camera
from toroidal_ray_tracing_amd import distributed as trtd
from toroidal_ray_tracing_amd.tracer import Tracer

ap = argparse.ArgumentParser()
ap.add_argument("--parts", type=int, default=8)
ap.add_argument("--streams", type=int, default=4)
ap.add_argument("--frames", type=int, default=384)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("specs", nargs="+")
a = ap.parse_args()
dev = torch.device("cuda:0")
W = H = 4096
sc, g, pc = camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(5)
K, n = a.streams, a.parts
G = trtd.default_group_rows(H, n, trtd.DEFAULT_CYCLES)
cur = torch.cuda.current_stream()
trs = [Tracer(0) for _ in range(K)]
frame = trtd.TiledFrame(trs, W, H, n, 0, dev, want_hits=("t", "px", "py", "pz", "nx", "ny", "nz"), gather="none", group_rows=G)
F = a.frames - a.frames % K


def apply(spec):
    keys = []
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("=")
        os.environ[k] = v
        keys.append(k)
    for t in trs:
        _tuning.reload(t)
    return keys


def batch():
    for _ in range(F):
        frame.render(sc, g, pc, abi.TRT_CAMERA_PINHOLE, cur)
    frame.join(cur)


res = {s: [] for s in a.specs}
for r in range(a.rounds + 1):
    for spec in a.specs:
        keys = apply(spec)
        batch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        batch()
        torch.cuda.synchronize()
        if r:
            res[spec].append((time.perf_counter() - t0) / F * 1e6)
        for k in keys:
            os.environ.pop(k)
for spec in a.specs:
    print(f"{n} parts, {K} streams, [{spec or 'defaults'}]: {statistics.median(res[spec]):.1f} us per frame (min {min(res[spec]):.1f})", flush=True)
