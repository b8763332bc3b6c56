#!/usr/bin/env python3
"""Small frames are launch-bound (two launches per frame): eager enqueue vs a captured hipGraph of 32 frames."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toroidal_ray_tracing_amd import camera
from toroidal_ray_tracing_amd.tracer import Tracer
dev = torch.device("cuda:0"); tr = Tracer(0); cur = torch.cuda.current_stream()
for W in (256, 512, 1024, 4096):
    H = W
    sc, g, pc = camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(5)
    img = torch.empty(H, W, 4, device=dev)
    F = 32
    for _ in range(3): tr.render_dev(sc, g, pc, W, H, img.data_ptr(), stream=cur.cuda_stream)
    torch.cuda.synchronize(); print(W, 'warm', flush=True)
    def eager():
        for _ in range(F): tr.render_dev(sc, g, pc, W, H, img.data_ptr(), stream=cur.cuda_stream)
    graph = torch.cuda.CUDAGraph(); side = torch.cuda.Stream(); side.wait_stream(cur)
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for _ in range(F): tr.render_dev(sc, g, pc, W, H, img.data_ptr(), stream=side.cuda_stream)
    cur.wait_stream(side); print(W, 'captured', flush=True)
    def t(fn):
        res = []
        for k in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
            if k: res.append((time.perf_counter() - t0) / F * 1e3)
        return statistics.median(res)
    print(f"{W}x{H}: eager {t(eager):.4f} ms/frame   hipGraph replay {t(graph.replay):.4f} ms/frame", flush=True)
