#!/usr/bin/env python3
"""The post pass (trt_post_dev -> rgba8) on the config-3 frame (85 % of its pixels grey: one pow per pixel) and on a random image
(three).  usage: post_frame.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toroidal_ray_tracing_amd import camera
from toroidal_ray_tracing_amd.tracer import Tracer
dev=torch.device('cuda:0'); tr=Tracer(0); s=torch.cuda.current_stream()
W=4096
rgba=torch.empty(W,W,4,device=dev); o8=torch.empty(W,W,4,dtype=torch.uint8,device=dev)
tr.render_dev(camera.single_torus_scene(), camera.baseline_camera(W,W), camera.baseline_push(5), W, W, rgba.data_ptr(), stream=s.cuda_stream)
rnd=torch.rand(W,W,4,device=dev); rnd[...,3]=1
def t(img):
    out=[]
    for k in range(6):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(20): tr.post_dev(img.data_ptr(), W*W, 0, o8.data_ptr(), stream=s.cuda_stream)
        e1.record(s); torch.cuda.synchronize()
        if k: out.append(e0.elapsed_time(e1)/20)
    return statistics.median(out)
for r in range(2):
    a=t(rgba); b=t(rnd)
    print(f"post -> rgba8, 4096^2: the config-3 frame {a:.4f} ms ({20*W*W/a/1e6:.0f} GB/s = {20*W*W/a/8e9*100:.1f} % of 8 TB/s); a random image {b:.4f} ms ({20*W*W/b/8e9*100:.1f} %)")
