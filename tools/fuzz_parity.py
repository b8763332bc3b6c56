#!/usr/bin/env python3
"""Long differential run of the GPU path against the CPU oracle (GPU box): N seeded random scenes / cameras / sizes /
variants / solvers / classification levels through ONE context, first-hit records bit for bit, query counts equal,
colours within the tolerance of tests/ (the tests run 24 + 40 such frames; this is the same check for as long as asked).
usage: fuzz_parity.py [N=400] [first_seed=100]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle   # the checker, never the thing measured
from toroidal_ray_tracing_amd import abi
from toroidal_ray_tracing_amd.tracer import Tracer
from test_gpu_parity import random_case, assert_hits_equal, q, COLOR_RTOL, COLOR_ATOL

n, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 400), (int(sys.argv[2]) if len(sys.argv) > 2 else 100)
oracle.lib()
t = Tracer(0)
rng = np.random.default_rng(first)
t0, bad = time.time(), 0
for k in range(n):
    sc, g, pc, W, H, cam = random_case(first + k)
    variant = ["listed", "persistent", "static"][int(rng.integers(0, 3))]
    solver = [abi.TRT_SOLVE_F32, abi.TRT_SOLVE_F64, abi.TRT_SOLVE_FERRARI_F32][int(rng.integers(0, 3))]
    if variant == "persistent" and solver == abi.TRT_SOLVE_FERRARI_F32:
        solver = abi.TRT_SOLVE_F32
    counted = bool(rng.integers(0, 2))   # the uncounted listed frames of multi-torus scenes take part in the cost feedback (stale costs
    t.set_render_variant(variant); t.set_solver(solver); t.set_classification(int(rng.integers(-1, 2))); t.enable_stats(counted)   # of OTHER scenes)
    rgba, hits = t.render(sc, g, pc, W, H, cam)
    if k % 3 == 0:   # the same frame again: now with ITS OWN history, if it is one that keeps one
        rgba, hits = t.render(sc, g, pc, W, H, cam)
    st = t.stats() if counted else None
    wr, wh, _, wst = oracle.render(sc, g, pc, W, H, cam, precision=solver, nthreads=16)
    try:
        assert_hits_equal(hits, wh, f"seed {first + k} ({variant}, solver {solver}, {W}x{H}, cam {cam})")
        np.testing.assert_allclose(rgba, wr, rtol=COLOR_RTOL, atol=COLOR_ATOL)
        assert st is None or q(st) == q(wst), (q(st), q(wst))
    except AssertionError as e:
        bad += 1
        print(f"MISMATCH seed {first + k}: {str(e)[:300]}", flush=True)
    if k % 50 == 49:
        print(f"{k + 1} frames, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
t.close()
print(f"fuzz: {n} frames, {bad} mismatches")
sys.exit(1 if bad else 0)
