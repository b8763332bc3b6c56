import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toroidal_ray_tracing_amd import abi, camera
from toroidal_ray_tracing_amd.tracer import Tracer
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
W = H = 4096
rgba = torch.empty(H, W, 4, device=dev)
hits = {k: torch.empty(H * W, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
hp = {k: v.data_ptr() for k, v in hits.items()}
def t(sc, g, pc, cam, solver, n=10, rounds=5):
    tr.set_solver(solver)
    out = []
    for r in range(rounds + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(n): tr.render_dev(sc, g, pc, W, H, rgba.data_ptr(), camera=cam, hit_ptrs=hp, stream=s.cuda_stream)
        e1.record(s); torch.cuda.synchronize()
        if r: out.append(e0.elapsed_time(e1) / n)
    tr.set_solver(0)
    return statistics.median(out)
names = {0: "walk32", 1: "walk64", 2: "dk32", 3: "dk64", 4: "ferrari32", 5: "ferrari64"}
pc = camera.baseline_push(5); pct = camera.baseline_push(5); pct.rho = 4.0
cases = [("single", camera.single_torus_scene(), camera.baseline_camera(W, H), pc, 0),
         ("nested8", camera.nested_tori_scene(), camera.baseline_camera(W, H), pc, 0),
         ("toroidal interior", camera.single_torus_scene(R=6.0, r=1.5, material=camera.PLASTIC), camera.toroidal_camera(W, H), pct, 1)]
for name, sc, g, p, cam in cases:
    print(name, "  ".join(f"{names[k]} {t(sc, g, p, cam, k):.4f}" for k in (0, 4, 1, 5, 2)))
