#!/usr/bin/env python3
"""Times every BASELINE config on one GPU (kernel time with HIP events, median of rounds).
Not the driver's benchmark (that is bench.py = config 3): a table for DESIGN.md."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toroidal_ray_tracing_amd import abi, camera
from toroidal_ray_tracing_amd.tracer import Tracer

dev = torch.device("cuda:0")
tr = Tracer(0)
s = torch.cuda.current_stream()


def timeit(fn, rounds=7, frames=10):
    for _ in range(3):
        fn()
    out = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(frames):
            fn()
        e1.record(s)
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / frames)
    return statistics.median(out)


def render_case(name, sc, g, pc, W, H, cam=0, f64=False, variants=("listed", "persistent", "static"), solver=None):
    rgba = torch.empty(H, W, 4, device=dev)
    hits = {k: torch.empty(H * W, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
    hp = {k: v.data_ptr() for k, v in hits.items()}
    tr.set_solver(solver if solver is not None else abi.TRT_SOLVE_F64 if f64 else abi.TRT_SOLVE_F32)
    for v in variants:
        tr.set_render_variant(v)
        tr.enable_stats(True)
        tr.render_dev(sc, g, pc, W, H, rgba.data_ptr(), camera=cam, hit_ptrs=hp, stream=s.cuda_stream)
        st = tr.stats()
        tr.enable_stats(False)
        ms = timeit(lambda: tr.render_dev(sc, g, pc, W, H, rgba.data_ptr(), camera=cam, hit_ptrs=hp, stream=s.cuda_stream))
        tot = st["primary_tests"] + st["bounce_tests"] + st["shadow_tests"]
        print(f"{name:34s} {v:10s} {ms:8.4f} ms  primary {st['primary_tests'] / ms / 1e6:8.1f} Gtests/s  "
              f"all {tot / ms / 1e6:8.1f} Gtests/s  {44 * W * H / ms / 1e6:7.0f} GB/s ({44 * W * H / ms / 8e9 * 100:4.1f}% of 8 TB/s)  "
              f"hit px {100 * (1 - 0):.0f}%" if False else
              f"{name:34s} {v:10s} {ms:8.4f} ms  primary {st['primary_tests'] / ms / 1e6:8.1f} Gtests/s  "
              f"all {tot / ms / 1e6:8.1f} Gtests/s  {44 * W * H / ms / 1e6:7.0f} GB/s ({44 * W * H / ms / 8e9 * 100:4.1f}% of 8 TB/s)")
    tr.set_solver(abi.TRT_SOLVE_F32)
    tr.set_render_variant("listed")


def trace_case(name, sc, g, pc, W, H, f64=False):
    """config 2: trace(rays_in -> hits_out) on the frame's primary rays (exported by a render)."""
    rend = torch.empty(W * H, 16, device=dev)
    tr.render_dev(sc, g, pc, W, H, 0, rendered_ptr=rend.data_ptr(), stream=s.cuda_stream)
    r = rend.view(W, H, 16).permute(1, 0, 2).reshape(-1, 16)
    rays = [r[:, 8 + k].contiguous() for k in range(3)] + [r[:, 12 + k].contiguous() for k in range(3)]
    out = {k: torch.empty(H * W, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
    op = {k: v.data_ptr() for k, v in out.items()}
    rp = [a.data_ptr() for a in rays]
    tr.set_solver(abi.TRT_SOLVE_F64 if f64 else abi.TRT_SOLVE_F32)
    ms = timeit(lambda: tr.trace_dev(sc, rp, W * H, op, stream=s.cuda_stream))
    tr.set_solver(abi.TRT_SOLVE_F32)
    hitf = torch.isfinite(out["t"]).float().mean().item()
    print(f"{name:34s} {'trace':10s} {ms:8.4f} ms  {W * H * sc.n_tori / ms / 1e6:8.1f} Gtests/s  {52 * W * H / ms / 1e6:7.0f} GB/s "
          f"({52 * W * H / ms / 8e9 * 100:4.1f}% of 8 TB/s)  hit fraction {hitf:.3f}")
    # dense case: the same number of rays, all aimed at the torus
    n = W * H
    gen = torch.Generator(device=dev).manual_seed(1)
    o = (torch.rand(n, 3, device=dev, generator=gen) * 8 - 4)
    tgt = torch.randn(n, 3, device=dev, generator=gen)
    tgt = tgt / tgt.norm(dim=1, keepdim=True) * (torch.rand(n, 1, device=dev, generator=gen) * 1.2)
    d = tgt - o
    d = d / d.norm(dim=1, keepdim=True)
    rays = [o[:, k].contiguous() for k in range(3)] + [d[:, k].contiguous() for k in range(3)]
    rp = [a.data_ptr() for a in rays]
    ms = timeit(lambda: tr.trace_dev(sc, rp, n, op, stream=s.cuda_stream))
    hitf = torch.isfinite(out["t"]).float().mean().item()
    print(f"{name + ' (random aimed rays)':34s} {'trace':10s} {ms:8.4f} ms  {n * sc.n_tori / ms / 1e6:8.1f} Gtests/s  {52 * n / ms / 1e6:7.0f} GB/s "
          f"({52 * n / ms / 8e9 * 100:4.1f}% of 8 TB/s)  hit fraction {hitf:.3f}")


W = 2048
trace_case("C2 2048^2 primary, 0 bounces", camera.single_torus_scene(), camera.baseline_camera(W, W), camera.baseline_push(1), W, W)
render_case("C2 as render, 2048^2 maxDepth 1", camera.single_torus_scene(), camera.baseline_camera(W, W), camera.baseline_push(1), W, W)
W = 4096
render_case("C3 4096^2 maxDepth 5", camera.single_torus_scene(), camera.baseline_camera(W, W), camera.baseline_push(5), W, W)
render_case("C3 with Durand-Kerner FP32", camera.single_torus_scene(), camera.baseline_camera(W, W), camera.baseline_push(5), W, W,
            variants=("listed",), solver=abi.TRT_SOLVE_DK_F32)
render_case("C3 with Durand-Kerner FP64", camera.single_torus_scene(), camera.baseline_camera(W, W), camera.baseline_push(5), W, W,
            variants=("listed",), solver=abi.TRT_SOLVE_DK_F64)
render_case("C3 with Ferrari FP32", camera.single_torus_scene(), camera.baseline_camera(W, W), camera.baseline_push(5), W, W,
            variants=("listed",), solver=abi.TRT_SOLVE_FERRARI_F32)
render_case("C3 with Ferrari FP64", camera.single_torus_scene(), camera.baseline_camera(W, W), camera.baseline_push(5), W, W,
            variants=("listed",), solver=abi.TRT_SOLVE_FERRARI_F64)
render_case("C4 8 nested tori, FP64 solve", camera.nested_tori_scene(), camera.baseline_camera(W, W), camera.baseline_push(5), W, W, f64=True)
render_case("C4' 8 nested tori, FP32 solve", camera.nested_tori_scene(), camera.baseline_camera(W, W), camera.baseline_push(5), W, W)
pc = camera.baseline_push(5); pc.rho = 4.0
render_case("toroidal camera, interior R=6", camera.single_torus_scene(R=6.0, r=1.5, material=camera.PLASTIC),
            camera.toroidal_camera(W, W), pc, W, W, cam=1)
n = 4096 * 4096
img = torch.rand(n, 4, device=dev)
img[:, 3] = 1.0   # a rendered frame: alpha is 1 (rgen:87)
o8 = torch.empty(n, 4, dtype=torch.uint8, device=dev)
of = torch.empty(n, 4, device=dev)
ms = timeit(lambda: tr.post_dev(img.data_ptr(), n, 0, o8.data_ptr(), stream=s.cuda_stream))
print(f"{'post pass 4096^2 -> unorm8':34s} {'post':10s} {ms:8.4f} ms  {20 * n / ms / 1e6:7.0f} GB/s ({20 * n / ms / 8e9 * 100:4.1f}% of 8 TB/s, 20 B/px)")
ms = timeit(lambda: tr.post_dev(img.data_ptr(), n, of.data_ptr(), 0, stream=s.cuda_stream))
print(f"{'post pass 4096^2 -> f32':34s} {'post':10s} {ms:8.4f} ms  {32 * n / ms / 1e6:7.0f} GB/s ({32 * n / ms / 8e9 * 100:4.1f}% of 8 TB/s, 32 B/px)")
# re-projection of a 4096x2048 capture (8.4 M points) into a 2048^2 view
n = 4096 * 2048
gen = torch.Generator(device=dev).manual_seed(2)
cloud = torch.zeros(n, 8, device=dev)
cloud[:, :3] = torch.rand(n, 3, device=dev, generator=gen) * 6 - 3
cloud[:, 4:7] = torch.rand(n, 3, device=dev, generator=gen)
vp = camera.perspective_vk(60, 1.0) @ camera.look_at((1.0, 2.0, 7.0), (0.0, 0.0, 0.0))
img2 = torch.empty(2048, 2048, 4, device=dev)
ms = timeit(lambda: tr.splat_dev(cloud.data_ptr(), n, vp, 2048, 2048, img2.data_ptr(), stream=s.cuda_stream))
print(f"{'re-projection 8.4M pts -> 2048^2':34s} {'splat':10s} {ms:8.4f} ms  {n / ms / 1e6:7.2f} Gpoints/s  "
      f"{(32 * n + 24 * 2048 * 2048) / ms / 1e6:7.0f} GB/s algorithmic (32 B/point + 24 B/pixel)")
W = 8192
render_case("C5 shape on ONE GPU: 8192^2", camera.single_torus_scene(), camera.baseline_camera(W, W), camera.baseline_push(5), W, W, variants=("listed",))
