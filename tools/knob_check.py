#!/usr/bin/env python3
"""Checks that experiment knobs (TRT_DEBUG_SKIP bits >= 8, other env) leave the frame bit-identical.
usage: knob_check.py ENV=VAL[,ENV=VAL] ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402  (loads the -DTRT_TUNING build, see _tuning.py)
import torch
from toroidal_ray_tracing_amd import abi, camera
from toroidal_ray_tracing_amd.tracer import Tracer

W = H = 1024
dev = torch.device("cuda:0")
tr = Tracer(0)
s = torch.cuda.current_stream()


def frame(sc, g, pc, cam):
    rgba = torch.full((H, W, 4), -7.0, device=dev)
    hits = {k: torch.full((H * W,), -7.0, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
    tr.render_dev(sc, g, pc, W, H, rgba.data_ptr(), camera=cam, hit_ptrs={k: v.data_ptr() for k, v in hits.items()},
                  stream=s.cuda_stream)
    torch.cuda.synchronize()
    return [rgba] + [hits[k] for k in sorted(hits)]


cases = [("single/pinhole", camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(5), 0),
         ("nested/pinhole", camera.nested_tori_scene(), camera.baseline_camera(W, H), camera.baseline_push(5), 0)]
bad = 0
for name, sc, g, pc, cam in cases:
    ref = frame(sc, g, pc, cam)
    for spec in sys.argv[1:]:
        kv = dict(x.split("=") for x in spec.split(","))
        os.environ.update(kv)
        _tuning.reload(tr)
        got = frame(sc, g, pc, cam)
        for k in kv:
            os.environ.pop(k)
        _tuning.reload(tr)
        same = all(torch.equal(a.view(torch.int32), b.view(torch.int32)) for a, b in zip(ref, got))
        print(f"{name:16s} {spec:32s} {'identical' if same else 'DIFFERENT'}")
        bad += not same
sys.exit(1 if bad else 0)
