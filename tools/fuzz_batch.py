#!/usr/bin/env python3
"""Long differential run of trt_render_batch_dev against the CPU oracle (GPU box): N seeded random batches — a random scene
(1-8 intersecting tori, every material), a random size, 1-8 frames each with its own camera position / field of view /
light / clear colour / maxDepth (pinhole) or its own rho / light / maxDepth (toroidal: the frames of a batch share eye and
centre), whole frames or the part of a random tiling, FP32 or FP64 solve, every classification level, counted or not —
through ONE context: every frame's first-hit record bit for bit, colours within the tolerance of tests/, the summed query
counts equal.  usage: fuzz_batch.py [N=300] [first_seed=5000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from oracle import oracle   # the checker, never the thing measured
from toroidal_ray_tracing_amd import abi, camera
from toroidal_ray_tracing_amd import distributed as trtd
from toroidal_ray_tracing_amd.tracer import Tracer
from test_gpu_parity import random_case, q, COLOR_RTOL, COLOR_ATOL, GEOM

n, first = (int(sys.argv[1]) if len(sys.argv) > 1 else 300), (int(sys.argv[2]) if len(sys.argv) > 2 else 5000)
oracle.lib()
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
t = Tracer(0)
t0, bad, frames_done, refused = time.time(), 0, 0, 0
for k in range(n):
    rng = np.random.default_rng(first + k)
    sc, g0, pc0, W, H, cam = random_case(first + k)
    B = int(rng.integers(1, abi.TRT_MAX_BATCH + 1))
    frames = []
    for j in range(B):
        _, gj, pcj, _, _, _ = random_case(first + k + 7919 * (j + 1))   # another camera, light, depth, rho …
        if cam == 1 or j == 0:
            gj = g0                                                      # toroidal: shared eye / centre (one set of tables) …
            if cam == 1 and j and k % 2:
                pcj.rho = pc0.rho                                        # … and rho: it enters theta when the eye is off the centre's height
        else:
            eye = rng.normal(size=3)
            gj = camera.globals_for(tuple(eye / np.linalg.norm(eye) * rng.uniform(2.0, 8.0)), tuple(rng.uniform(-1, 1, 3)), W, H,
                                    fov_deg=float(rng.uniform(20.0, 110.0)))
        frames.append((gj, pcj if j else pc0))
    tiling = None
    if rng.integers(0, 2):
        parts = int(rng.integers(2, 5))
        tiling = abi.trt_tiling(8 * int(rng.integers(1, 3)), parts, int(rng.integers(0, parts)), int(rng.integers(0, 2)))
    solver = [abi.TRT_SOLVE_F32, abi.TRT_SOLVE_F64][int(rng.integers(0, 2))]
    counted = bool(rng.integers(0, 2))
    t.set_solver(solver); t.set_classification(int(rng.integers(-1, 2))); t.enable_stats(counted)
    compact = tiling is not None and tiling.compact
    rows_idx = np.array(trtd.owned_rows(H, tiling.group_rows, tiling.n_parts, tiling.part), dtype=np.int64) if tiling is not None else np.arange(H)
    rows = len(rows_idx) if compact else H
    rows = max(rows, 1)   # (a part may own no row of a small frame: the call then launches nothing)
    outs = []
    for _ in frames:
        rgba = torch.zeros(rows, W, 4, device=dev)
        hits = {kk: torch.zeros(rows * W, device=dev) for kk in GEOM}
        hits["id"] = torch.zeros(rows * W, dtype=torch.int32, device=dev)
        outs.append((rgba, hits))
    fl = [(g, pc, o[0].data_ptr(), {kk: v.data_ptr() for kk, v in o[1].items()}) for (g, pc), o in zip(frames, outs)]
    try:
        for _ in range(2 if k % 3 == 0 else 1):   # every third batch twice: the second time with its own cost history
            t.render_batch_dev(sc, fl, W, H, tiling, camera=cam, stream=s)
    except Exception as e:
        # toroidal frames that differ in rho with the eye above / below the centre have different tables: refused, by contract
        if cam == 1 and "one by one" in str(e) and any(f[1].rho != pc0.rho for f in frames):
            refused += 1
            continue
        raise
    torch.cuda.synchronize()
    st = t.stats() if counted else None
    want_q = {kk: 0 for kk in q({kk: 0 for kk in abi.STAT_FIELDS})}
    try:
        for j, ((g, pc), (rgba, hits)) in enumerate(zip(frames, outs)):
            wr, wh, _, wst = oracle.render(sc, g, pc, W, H, cam, precision=solver, nthreads=16)
            sel = rows_idx if tiling is not None else slice(None)
            got_rgba = rgba.cpu().numpy()
            got_rgba = got_rgba[:len(rows_idx)] if compact else (got_rgba if tiling is None else got_rgba[rows_idx])
            np.testing.assert_allclose(got_rgba, wr[sel], rtol=COLOR_RTOL, atol=COLOR_ATOL, err_msg=f"frame {j} colours")
            for kk in GEOM + ("id",):
                a = hits[kk].cpu().numpy().reshape(rows, W)
                a = a[:len(rows_idx)] if compact else (a if tiling is None else a[rows_idx])
                b = wh[kk].reshape(H, W)[sel]
                assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"frame {j} of {B}: {kk} bits"
            if tiling is None:
                for kk in want_q:
                    want_q[kk] += wst[kk]
        if st is not None and tiling is None:
            assert q(st) == want_q, (q(st), want_q)
    except AssertionError as e:
        bad += 1
        print(f"MISMATCH seed {first + k} ({B} frames, {W}x{H}, cam {cam}, solver {solver}, tiling {None if tiling is None else (tiling.group_rows, tiling.n_parts, tiling.part, tiling.compact)}): {str(e)[:300]}", flush=True)
    frames_done += B
    if k % 50 == 49:
        print(f"{k + 1} batches ({frames_done} frames), {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
t.close()
print(f"fuzz_batch: {n} batches ({frames_done} frames rendered, {refused} toroidal batches with differing tables refused as documented), {bad} mismatches")
sys.exit(1 if bad else 0)
