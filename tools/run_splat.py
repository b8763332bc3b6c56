#!/usr/bin/env python3
"""Driver for profiling trt_splat_dev under rocprofv3: N launches, 8.4 M random points -> 2048²."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402
import torch
from toroidal_ray_tracing_amd import camera
from toroidal_ray_tracing_amd.tracer import Tracer
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
n = 4096 * 2048
gen = torch.Generator(device=dev).manual_seed(2)
cloud = torch.zeros(n, 8, device=dev)
cloud[:, :3] = torch.rand(n, 3, device=dev, generator=gen) * 6 - 3
cloud[:, 4:7] = torch.rand(n, 3, device=dev, generator=gen)
vp = camera.perspective_vk(60, 1.0) @ camera.look_at((1.0, 2.0, 7.0), (0.0, 0.0, 0.0))
img = torch.empty(2048, 2048, 4, device=dev)
for _ in range(reps):
    tr.splat_dev(cloud.data_ptr(), n, vp, 2048, 2048, img.data_ptr(), stream=s.cuda_stream)
torch.cuda.synchronize()
