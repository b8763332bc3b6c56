#!/usr/bin/env python3
"""Per-wave timeline of the listed render kernel (GPU box).  Needs the -DTRT_TIMELINE build:
    make -C toroidal_ray_tracing_amd/csrc timeline        (-> toroidal_ray_tracing_amd/libtrt_timeline.so)
Every wave stamps the 100-MHz wall clock at entry (t0), before its first tile (t1) and at exit (t2) plus HW_ID / XCC_ID.
usage: timeline.py [c3|c3live|c4|c4f32|toro|capture]   — c3live = config 3 with the CLEAR tiles skipped (TRT_DEBUG_SKIP=1)
Prints: kernel span, ramp (first→last wave start), waves in flight over time, the prologue and tile time distributions,
and what the kernel would take if the same wave-seconds were spread evenly over the resident slots."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("TRT_LIB", os.path.join(ROOT, "toroidal_ray_tracing_amd", "libtrt_timeline.so"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _tuning  # noqa: E402
import numpy as np
import torch
from toroidal_ray_tracing_amd import abi, camera
from toroidal_ray_tracing_amd.tracer import Tracer

case = sys.argv[1] if len(sys.argv) > 1 else "c3live"
dev = torch.device("cuda:0"); tr = Tracer(0); s = torch.cuda.current_stream()
fn = tr._L.trt_debug_set_timeline; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
W = H = 4096
sc, g, pc, cam = camera.single_torus_scene(), camera.baseline_camera(W, H), camera.baseline_push(5), abi.TRT_CAMERA_PINHOLE
if case == "c3live":
    os.environ["TRT_DEBUG_SKIP"] = "1"; _tuning.reload(tr)
elif case in ("c4", "c4f32"):
    sc = camera.nested_tori_scene()
    if case == "c4": tr.set_solver(abi.TRT_SOLVE_F64)
rend = None
if case in ("toro", "capture"):   # the toroidal capture of bench.py / quick.py (4096 x 2048, camera inside an R=6 torus); capture: + RenderedData
    W, H = 4096, 2048
    sc = camera.single_torus_scene(R=6.0, r=1.5, material=camera.PLASTIC)
    g, cam = camera.toroidal_camera(W, H), abi.TRT_CAMERA_TOROIDAL
    pc.rho = 4.0
    if case == "capture":
        rend = torch.empty(W * H, 16, device=dev)
rgba = torch.empty(H, W, 4, device=dev)
hits = {k: torch.empty(W * H, device=dev) for k in ("t", "px", "py", "pz", "nx", "ny", "nz")}
hp = {k: v.data_ptr() for k, v in hits.items()}
frame = lambda: tr.render_dev(sc, g, pc, W, H, rgba.data_ptr(), camera=cam, hit_ptrs=hp, stream=s.cuda_stream,
                              rendered_ptr=rend.data_ptr() if rend is not None else 0)
for _ in range(20): frame()
torch.cuda.synchronize()
NW = 1 << 20   # more than any grid has waves
buf = torch.zeros(NW, 8, dtype=torch.int64, device=dev)
assert fn(tr._h, buf.data_ptr()) == 0
for _ in range(3):
    buf.zero_(); frame(); torch.cuda.synchronize()
assert fn(tr._h, None) == 0
t = buf.cpu().numpy()
used = t[:, 0] != 0
t = t[used]; n = len(t)
t0, t1, t2, hw = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
base = t0.min(); tick = 0.01   # µs per tick (100 MHz)
span = (t2.max() - base) * tick
print(f"{case}: {n} waves stamped, {(t[:, 5] == 0).sum()} of them left before staging (their block owns no list entry); kernel span (first wave in → last wave out) {span:.1f} us; last wave start at {(t0.max() - base) * tick:.1f} us")
has_tile = t1 != 0
pro = (np.where(has_tile, t1, t2) - t0) * tick
body = (t2 - np.where(has_tile, t1, t2)) * tick
stg = ((t[:, 5] - t0) * tick)[t[:, 5] != 0]   # (blocks that own no list entry leave before the staging)
for nm, v in (("staging (entry → past the barrier)", stg), ("prologue (entry → lists settled)", pro), ("tiles (→ exit)", body[has_tile]), ("whole wave", (t2 - t0) * tick)):
    q = np.percentile(v, [10, 50, 90, 99, 100])
    print(f"  {nm:36s} mean {v.mean():6.2f}  p10 {q[0]:6.2f}  p50 {q[1]:6.2f}  p90 {q[2]:6.2f}  p99 {q[3]:6.2f}  max {q[4]:6.2f} us   (sum {v.sum() / 1e3:.1f} wave-ms)")
# waves in flight over time
edges = np.linspace(0, span, 41)
mid = 0.5 * (edges[1:] + edges[:-1])
inflight = [(((t0 - base) * tick <= m) & ((t2 - base) * tick > m)).sum() for m in mid]
work = [(((np.where(has_tile, t1, t2) - base) * tick <= m) & ((t2 - base) * tick > m)).sum() for m in mid]
print("  waves in flight (and of those past their prologue) at 40 points of the span; 1024 SIMDs:")
print("   " + " ".join(f"{a}" for a in inflight))
print("   " + " ".join(f"{a}" for a in work))
ws = ((t2 - t0) * tick).sum()
slots = 1024 * 6
print(f"  wave-seconds {ws / 1e3:.1f} wave-ms over {slots} slots = {ws / slots:.1f} us if evenly packed (span {span:.1f} us)")
# per-XCC and per-CU balance: sum of wave time per SIMD
xcc = (hw >> 32) & 0xf
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3; sh = (hw >> 12) & 1
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
busy = np.bincount(key * 4 + simd, weights=(t2 - t0) * tick)
busy = busy[busy > 0]
print(f"  SIMDs seen {len(busy)}; wave-time per SIMD: mean {busy.mean():.1f} us, min {busy.min():.1f}, max {busy.max():.1f}  (÷ span = mean resident waves {busy.mean() / span:.2f})")
bx = np.bincount(xcc, weights=(t2 - t0) * tick)
print("  wave-time per XCC (wave-ms): " + " ".join(f"{v / 1e3:.1f}" for v in bx))
last = np.array([((t2 - base) * tick)[xcc == k].max() for k in range(int(xcc.max()) + 1)])
print("  last wave out per XCC (us): " + " ".join(f"{v:.1f}" for v in last))
# where the heavy tiles are: mean / max tile time on a 32 x 32 map of the image (rows = y)
tile = t[:, 4]
ok = (tile >> 32) == 1
tx, ty, miss = (tile & 0xffff)[ok], ((tile >> 16) & 0x7fff)[ok], ((tile >> 31) & 1)[ok]
bt = body[ok]
print(f"  LIVE tiles {ok.sum()} ({miss.sum()} of them flagged miss); tile time by start order: first quarter mean {bt[: len(bt) // 4].mean():.1f}, last quarter {bt[-len(bt) // 4:].mean():.1f} us")
G = 32
cell = (ty * 8 * G // H) * G + (tx * 8 * G // W)
mx = np.zeros(G * G); np.maximum.at(mx, cell, bt)
sm = np.bincount(cell, weights=bt, minlength=G * G); cn = np.bincount(cell, minlength=G * G)
print("  max tile time (us) per 1/32 x 1/32 image cell, rows top to bottom ('.' = no LIVE tile):")
for r in range(G):
    if cn[r * G:(r + 1) * G].sum():
        print("   " + " ".join(f"{mx[r * G + c]:3.0f}" if cn[r * G + c] else "  ." for c in range(G)))
order = np.argsort(-bt)[:12]
print("  heaviest tiles (tx, ty, us, start us): " + ", ".join(f"({tx[i]},{ty[i]},{bt[i]:.0f},{((t0[ok][i]) - base) * tick:.0f})" for i in order))
