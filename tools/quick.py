#!/usr/bin/env python3
"""Quick table of the kernels under work (GPU box): C3 frame (+ LIVE / CLEAR parts alone), C4, trt_trace on camera
and aimed rays, the toroidal capture with RenderedData, the re-projection.  usage: quick.py [tag] [cases…]
A/B of two builds: run it twice with TRT_LIB=… (alternate the processes, cdna guide §5.4 rule 24)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402
import torch
from toroidal_ray_tracing_amd import abi, camera
from toroidal_ray_tracing_amd.tracer import Tracer

tag = sys.argv[1] if len(sys.argv) > 1 else ""
cases = set(sys.argv[2:]) or {"c3", "parts", "c4", "trace", "capture", "splat", "persist"}
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
V = os.environ.get("VARIANT", "listed")   # VARIANT=persistent|static: every render case with that variant
tr.set_render_variant(V)


def timeit(fn, rounds=7, reps=20):
    for _ in range(5):
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
    return statistics.median(out), min(out)


def show(name, ms, bytes_):
    print(f"{tag:10s} {name:44s} {ms[0]:8.4f} ms (min {ms[1]:.4f})  {bytes_ / ms[0] / 1e6:7.0f} GB/s  {bytes_ / ms[0] / 8e9 * 100:5.1f} %", flush=True)


W = 4096
rgba = torch.empty(W, W, 4, device=dev)
hits = {k: torch.empty(W * W, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
hp = {k: v.data_ptr() for k, v in hits.items()}
sc1, g, pc = camera.single_torus_scene(), camera.baseline_camera(W, W), camera.baseline_push(5)
if "c3" in cases:
    show("C3 4096^2 listed", timeit(lambda: tr.render_dev(sc1, g, pc, W, W, rgba.data_ptr(), hit_ptrs=hp, stream=s.cuda_stream)), 44 * W * W)
if "c3alt" in cases:
    # the same frame into FOUR alternating output sets (3 GB): nothing a frame writes can still sit in the 256-MB Infinity Cache
    # when the next frame writes the same addresses — what the memory side of the frame costs without that reuse
    sets = [(torch.empty(W, W, 4, device=dev), {k: torch.empty(W * W, device=dev) for k in hits}) for _ in range(4)]
    ptrs = [(r.data_ptr(), {k: v.data_ptr() for k, v in h.items()}) for r, h in sets]
    state = {"i": 0}
    def alt():
        r, h = ptrs[state["i"] & 3]; state["i"] += 1
        tr.render_dev(sc1, g, pc, W, W, r, hit_ptrs=h, stream=s.cuda_stream)
    show("C3 into 4 alternating output sets", timeit(alt), 44 * W * W)
    if "partsalt" in cases:   # the two halves of the frame alone, without the reuse
        for skip, nm in ((1, "C3 LIVE part alone, 4 alternating sets"), (2, "C3 CLEAR part alone, 4 alternating sets")):
            os.environ["TRT_DEBUG_SKIP"] = str(skip); _tuning.reload(tr)
            show(nm, timeit(alt), 44 * W * W)
        os.environ.pop("TRT_DEBUG_SKIP"); _tuning.reload(tr)
    del sets
if "parts" in cases:
    for skip, nm in ((1, "C3 LIVE part alone (CLEAR skipped)"), (2, "C3 CLEAR part alone (LIVE skipped)")):
        os.environ["TRT_DEBUG_SKIP"] = str(skip); _tuning.reload(tr)
        show(nm, timeit(lambda: tr.render_dev(sc1, g, pc, W, W, rgba.data_ptr(), hit_ptrs=hp, stream=s.cuda_stream)), 44 * W * W)
    os.environ.pop("TRT_DEBUG_SKIP"); _tuning.reload(tr)
if "c5" in cases:
    W5 = 8192
    rgba5 = torch.empty(W5, W5, 4, device=dev)
    hits5 = {k: torch.empty(W5 * W5, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
    hp5 = {k: v.data_ptr() for k, v in hits5.items()}
    g5 = camera.baseline_camera(W5, W5)
    show("C5 shape 8192^2 on one GPU", timeit(lambda: tr.render_dev(sc1, g5, pc, W5, W5, rgba5.data_ptr(), hit_ptrs=hp5, stream=s.cuda_stream), reps=10), 44 * W5 * W5)
    del rgba5, hits5
if "persist" in cases:
    tr.set_render_variant("persistent")
    show("C3 persistent variant", timeit(lambda: tr.render_dev(sc1, g, pc, W, W, rgba.data_ptr(), hit_ptrs=hp, stream=s.cuda_stream)), 44 * W * W)
    tr.set_render_variant(V)
if "c4" in cases:
    sc8 = camera.nested_tori_scene()
    tr.set_solver(abi.TRT_SOLVE_F64)
    show("C4 8 nested tori FP64", timeit(lambda: tr.render_dev(sc8, g, pc, W, W, rgba.data_ptr(), hit_ptrs=hp, stream=s.cuda_stream), reps=10), 44 * W * W)
    tr.set_solver(abi.TRT_SOLVE_F32)
    show("C4' 8 nested tori FP32", timeit(lambda: tr.render_dev(sc8, g, pc, W, W, rgba.data_ptr(), hit_ptrs=hp, stream=s.cuda_stream), reps=10), 44 * W * W)
if "trace" in cases:
    Wt = 2048; n = Wt * Wt
    g2, pc1 = camera.baseline_camera(Wt, Wt), camera.baseline_push(1)
    rend = torch.empty(n, 16, device=dev)
    tr.render_dev(sc1, g2, pc1, Wt, Wt, 0, rendered_ptr=rend.data_ptr(), stream=s.cuda_stream)
    r = rend.view(Wt, Wt, 16).permute(1, 0, 2).reshape(-1, 16)
    rays = [r[:, 8 + k].contiguous() for k in range(3)] + [r[:, 12 + k].contiguous() for k in range(3)]
    out = {k: torch.empty(n, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
    op = {k: v.data_ptr() for k, v in out.items()}
    rp = [a.data_ptr() for a in rays]
    show("trace 2048^2 camera rays (10 % hit)", timeit(lambda: tr.trace_dev(sc1, rp, n, op, stream=s.cuda_stream)), 52 * n)
    gen = torch.Generator(device=dev).manual_seed(1)
    o = torch.rand(n, 3, device=dev, generator=gen) * 8 - 4
    tgt = torch.randn(n, 3, device=dev, generator=gen)
    tgt = tgt / tgt.norm(dim=1, keepdim=True) * (torch.rand(n, 1, device=dev, generator=gen) * 1.2)
    d = tgt - o; d = d / d.norm(dim=1, keepdim=True)
    rays2 = [o[:, k].contiguous() for k in range(3)] + [d[:, k].contiguous() for k in range(3)]
    rp2 = [a.data_ptr() for a in rays2]
    show("trace 2048^2 aimed rays (45 % hit)", timeit(lambda: tr.trace_dev(sc1, rp2, n, op, stream=s.cuda_stream)), 52 * n)
    sc8 = camera.nested_tori_scene()
    show("trace 2048^2 aimed rays, 8 nested tori", timeit(lambda: tr.trace_dev(sc8, rp2, n, op, stream=s.cuda_stream), reps=10), 52 * n)
    del rend, r, rays, rays2
if "capture" in cases:
    Wc, Hc = 4096, 2048
    sct = camera.single_torus_scene(R=6.0, r=1.5, material=camera.PLASTIC)
    gt = camera.toroidal_camera(Wc, Hc)
    pct = camera.baseline_push(5); pct.rho = 4.0
    rend = torch.empty(Wc * Hc, 16, device=dev)
    show("toroidal 4096x2048, no RenderedData", timeit(lambda: tr.render_dev(sct, gt, pct, Wc, Hc, rgba.data_ptr(), camera=1, hit_ptrs=hp, stream=s.cuda_stream), reps=10), 44 * Wc * Hc)
    show("toroidal 4096x2048 + RenderedData (capture)", timeit(lambda: tr.render_dev(sct, gt, pct, Wc, Hc, rgba.data_ptr(), camera=1, hit_ptrs=hp,
                                                                                    rendered_ptr=rend.data_ptr(), stream=s.cuda_stream), reps=10), 108 * Wc * Hc)
    show("toroidal capture, RenderedData only", timeit(lambda: tr.render_dev(sct, gt, pct, Wc, Hc, 0, camera=1, rendered_ptr=rend.data_ptr(), stream=s.cuda_stream), reps=10), 64 * Wc * Hc)
    del rend
if "splat" in cases:
    n = 4096 * 2048
    gen = torch.Generator(device=dev).manual_seed(2)
    cloud = torch.zeros(n, 8, device=dev)
    cloud[:, :3] = torch.rand(n, 3, device=dev, generator=gen) * 6 - 3
    cloud[:, 4:7] = torch.rand(n, 3, device=dev, generator=gen)
    vp = camera.perspective_vk(60, 1.0) @ camera.look_at((1.0, 2.0, 7.0), (0.0, 0.0, 0.0))
    img2 = torch.empty(2048, 2048, 4, device=dev)
    show("re-projection 8.4M random pts -> 2048^2", timeit(lambda: tr.splat_dev(cloud.data_ptr(), n, vp, 2048, 2048, img2.data_ptr(), stream=s.cuda_stream), reps=5),
         32 * n + 24 * 2048 * 2048)
