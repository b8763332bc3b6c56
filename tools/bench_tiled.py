#!/usr/bin/env python3
"""What ONE rank of an N-GPU job renders per frame, measured part by part on one GPU: the 4096² (or --size) frame of
config 3 tiled over N parts with TiledFrame's default interleaving; prints every part's time per frame and what the
slowest part allows for the sharded renders alone (no collective) — the ceiling of `bench.py --gpus N`.
With --batch every part is rendered N frames per launch (trt_render_batch_dev, what bench.py --gpus N does): N parts of
1/N frame are the work of one full frame.
usage: bench_tiled.py [--size 4096] [--parts 2 4 8] [--cycles 8] [--batch]"""
import argparse, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toroidal_ray_tracing_amd import abi, camera
from toroidal_ray_tracing_amd import distributed as trtd
from toroidal_ray_tracing_amd.tracer import Tracer

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--parts", type=int, nargs="+", default=[1, 2, 4, 8])
ap.add_argument("--cycles", type=int, default=trtd.DEFAULT_CYCLES)
ap.add_argument("--batch", action="store_true", help="N frames per launch for the N-part tiling")
a = ap.parse_args()
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
W = H = a.size
sc, g, pc = camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(5)


def timeit(fn, rounds=5, reps=64):
    for _ in range(8):
        fn()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            fn()
        e1.record(s)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return statistics.median(out)


base = None
for n in a.parts:
    G = trtd.default_group_rows(H, n, a.cycles)
    times = []
    for part in range(n):
        t = abi.trt_tiling(G, n, part, 1 if n > 1 else 0)
        rows = tr.tiling_rows(t, H) if n > 1 else H
        rgba = torch.empty(rows, W, 4, device=dev)
        hits = {k: torch.empty(rows * W, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
        hp = {k: v.data_ptr() for k, v in hits.items()}
        if n == 1:
            times.append(timeit(lambda: tr.render_dev(sc, g, pc, W, H, rgba.data_ptr(), hit_ptrs=hp, stream=s.cuda_stream)))
        elif a.batch:
            sets = [(torch.empty(rows, W, 4, device=dev), {k: torch.empty(rows * W, device=dev) for k in hits}) for _ in range(n)]
            fl = [(g, pc, r.data_ptr(), {k: v.data_ptr() for k, v in h.items()}) for r, h in sets]
            times.append(timeit(lambda: tr.render_batch_dev(sc, fl, W, H, t, stream=s.cuda_stream), reps=16) / n)
            del sets
        else:
            times.append(timeit(lambda: tr.render_tiled_dev(sc, g, pc, W, H, t, rgba.data_ptr(), hit_ptrs=hp, stream=s.cuda_stream)))
        del rgba, hits
    worst = max(times)
    base = base or worst
    print(f"{n} part(s){' x ' + str(n) + ' frames per launch' if a.batch and n > 1 else ''}, groups of {G} rows: per part-frame " + " ".join(f"{x * 1e3:.1f}" for x in times) + f" us; slowest {worst * 1e3:.1f} us "
          f"-> {W * H / worst / 1e6:.1f} G primary tests/s for the sharded renders alone = {base / worst:.2f}x of one GPU "
          f"({base / worst / n * 100:.0f} % of linear)", flush=True)
