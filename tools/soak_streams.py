#!/usr/bin/env python3
"""Concurrency soak (GPU box): K contexts on K HIP streams render DIFFERENT frames at the same time, for N rounds; every
50 rounds each stream's output is compared bit for bit with the frame rendered alone.  What `bench.py --gpus N` relies
on with several frames in flight per rank: contexts share nothing.  usage: soak_streams.py [K=4] [rounds=2000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toroidal_ray_tracing_amd import abi, camera
from toroidal_ray_tracing_amd.tracer import Tracer

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device("cuda:0")
cases = [  # (scene, W, H, camera model, maxDepth, solver, tiling)
    (camera.single_torus_scene(), 4096, 4096, 0, 5, abi.TRT_SOLVE_F32, abi.trt_tiling(64, 8, 3, 1)),
    (camera.nested_tori_scene(), 1024, 1024, 0, 5, abi.TRT_SOLVE_F64, None),
    (camera.single_torus_scene(R=6.0, r=1.5, material=camera.PLASTIC), 2048, 1024, 1, 3, abi.TRT_SOLVE_F32, None),
    (camera.single_torus_scene(), 2048, 2048, 0, 1, abi.TRT_SOLVE_F32, abi.trt_tiling(256, 2, 1, 1)),
    (camera.nested_tori_scene(), 1536, 1000, 0, 4, abi.TRT_SOLVE_F32, None),
    (camera.single_torus_scene(), 777, 333, 0, 5, abi.TRT_SOLVE_FERRARI_F32, None),
][:K]
trs, streams, bufs, refs, calls = [], [], [], [], []
for sc, W, H, cam, depth, solver, tiling in cases:
    tr = Tracer(0)
    tr.set_solver(solver)
    g = camera.toroidal_camera(W, H) if cam else camera.baseline_camera(W, H)
    pc = camera.baseline_push(depth)
    if cam:
        pc.rho = 4.0
    rows = tr.tiling_rows(tiling, H) if tiling is not None else H
    out = torch.zeros(rows, W, 4, device=dev)
    hits = {k: torch.zeros(rows * W, device=dev) for k in ("t", "px", "ny")}
    hp = {k: v.data_ptr() for k, v in hits.items()}
    st = torch.cuda.Stream(device=dev)
    if tiling is not None:
        call = (lambda tr=tr, sc=sc, g=g, pc=pc, W=W, H=H, t=tiling, o=out, hp=hp, st=st, cam=cam:
                tr.render_tiled_dev(sc, g, pc, W, H, t, o.data_ptr(), camera=cam, hit_ptrs=hp, stream=st.cuda_stream))
    else:
        call = (lambda tr=tr, sc=sc, g=g, pc=pc, W=W, H=H, o=out, hp=hp, st=st, cam=cam:
                tr.render_dev(sc, g, pc, W, H, o.data_ptr(), camera=cam, hit_ptrs=hp, stream=st.cuda_stream))
    call()
    torch.cuda.synchronize()
    refs.append([out.clone()] + [hits[k].clone() for k in sorted(hits)])
    trs.append(tr); streams.append(st); bufs.append((out, hits)); calls.append(call)
bad, t0 = 0, time.time()
for r in range(rounds):
    if r % 50 == 0:
        for out, hits in bufs:
            out.zero_(); [v.zero_() for v in hits.values()]
        torch.cuda.synchronize()
    for c in calls:
        c()
    if r % 50 == 49:
        torch.cuda.synchronize()
        for i, (out, hits) in enumerate(bufs):
            got = [out] + [hits[k] for k in sorted(hits)]
            same = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(got, refs[i]))
            bad += not same
            if not same:
                print(f"round {r + 1}: stream {i} DIFFERS", flush=True)
    if r % 500 == 499:
        print(f"{r + 1} rounds x {len(calls)} concurrent frames, {bad} differences, {time.time() - t0:.0f} s", flush=True)
for tr in trs:
    tr.close()
print(f"soak_streams: {rounds} rounds x {len(calls)} streams, {bad} differences")
sys.exit(1 if bad else 0)
