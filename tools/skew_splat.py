#!/usr/bin/env python3
"""Re-projection on clouds that crowd a few bins — where the paged scatter's page ends and their waits are — against the
two-pass sorted scatter (TRT_SPLAT_VARIANT=1) and the one-pass form (=0), tuning build.  usage: skew_splat.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402
import torch
from toroidal_ray_tracing_amd import camera
from toroidal_ray_tracing_amd.tracer import Tracer
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
gen = torch.Generator(device=dev).manual_seed(3)
W = H = 2048
vp = camera.perspective_vk(60, 1.0) @ camera.look_at((0.0, 0.0, 5.0), (0.0, 0.0, 0.0))
img = torch.empty(H, W, 4, device=dev)
def cloud(n, spread):
    c = torch.zeros(n, 8, device=dev)
    c[:, :3] = (torch.rand(n, 3, device=dev, generator=gen) - 0.5) * spread
    c[:, 4:7] = torch.rand(n, 3, device=dev, generator=gen)
    return c
def t(c, n, reps=5, rounds=4):
    res = []
    for k in range(rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): tr.splat_dev(c.data_ptr(), n, vp, W, H, img.data_ptr(), stream=s.cuda_stream)
        e1.record(s); torch.cuda.synchronize()
        if k: res.append(e0.elapsed_time(e1) / reps)
    return statistics.median(res)
for n in (2_000_000, 8_388_608):
    for spread, name in ((6.0, "the whole view"), (2.0, "a third of the view"), (0.5, "a twelfth of the view (a few bins)"), (0.05, "one bin")):
        c = cloud(n, spread)
        row = []
        for v, nm in ((None, "paged"), ("1", "two-pass"), ("0", "one-pass")):
            if v is None: os.environ.pop("TRT_SPLAT_VARIANT", None)
            else: os.environ["TRT_SPLAT_VARIANT"] = v
            _tuning.reload(tr)
            row.append(f"{nm} {t(c, n):.3f} ms")
        os.environ.pop("TRT_SPLAT_VARIANT", None); _tuning.reload(tr)
        print(f"{n:>9d} points over {name}: " + ", ".join(row), flush=True)
        del c
