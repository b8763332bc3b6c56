import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from toroidal_ray_tracing_amd import camera
from toroidal_ray_tracing_amd.tracer import Tracer
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
W = H = 4096
sc, g, pc = camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(5)
a = torch.zeros(H, W, 4, device=dev); b = torch.zeros(H, W, 4, device=dev)
ha = {k: torch.zeros(W * H, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
hb = {k: torch.zeros(W * H, device=dev) for k in ha}
tr.render_dev(sc, g, pc, W, H, a.data_ptr(), hit_ptrs={k: v.data_ptr() for k, v in ha.items()}, stream=s.cuda_stream)
torch.cuda.synchronize()
t0 = time.perf_counter(); bad = 0
for k in range(3000):
    tr.render_dev(sc, g, pc, W, H, b.data_ptr(), hit_ptrs={k2: v.data_ptr() for k2, v in hb.items()}, stream=s.cuda_stream)
    if k % 500 == 499:
        torch.cuda.synchronize()
        same = torch.equal(a.view(torch.int32), b.view(torch.int32)) and all(torch.equal(ha[q].view(torch.int32), hb[q].view(torch.int32)) for q in ha)
        bad += not same
        print(k + 1, "frames", "identical" if same else "DIFFERENT", f"{(time.perf_counter() - t0) / (k + 1) * 1e3:.4f} ms/frame wall", flush=True)
        b.zero_(); [v.zero_() for v in hb.values()]
sys.exit(1 if bad else 0)
