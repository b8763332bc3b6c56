#!/usr/bin/env python3
"""What re-ordering rays could buy trt_trace on incoherent rays, from the REAL per-ray work of bench.py's
aimed set: every ray's walk length (polynomial evaluations, 0 = culled in setup) comes from the CPU oracle's
torus_first_hit, the sort keys a kernel could know after setup() (culled / split / pieces of the window / start direction)
from a numpy restatement of setup().  The kernel is VALU issue-bound (profiles/r02_pmc_trace_aimed.json: 92 % of the issue
slots), so its time is its wave-instruction count; a wave walks to its slowest lane.  Model per wave of 64 rays:
   fixed  F = load + setup + finish + normal + stores,   walk = C_TRIP x (evaluations of the slowest lane)
with C_TRIP calibrated so that today's order gives the measured 716 wave-instructions per 64 rays.  Organisations:
   today        64 consecutive rays per wave
   key(W)       the rays of a window of W rays (one wave's LDS list: 256..1024; 'all' = a global sort, i.e. a second launch)
                stably sorted by what setup() knows, culled rays dropped from the walk; +RESETUP instructions per 64 rays
                (setup() of the survivors is redone after the sort — parking their state costs more)
   ideal(W)     the same sorted by the TRUE walk length: the bound of ANY re-ordering inside a window
usage: sim_trace_sort.py [n_rays=131072]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
rng = np.random.default_rng(1)
# bench.py's aimed set: origins uniform in [-4,4]^3, targets uniform in direction with radius uniform in [0,1.2]
o = rng.uniform(-4, 4, (n, 3))
tgt = rng.normal(size=(n, 3))
tgt *= rng.uniform(0, 1.2, (n, 1)) / np.linalg.norm(tgt, axis=1, keepdims=True)
d = tgt - o
d /= np.linalg.norm(d, axis=1, keepdims=True)
o32, d32 = o.astype(np.float32), d.astype(np.float32)
R, r = 1.0, 0.25
ev = np.zeros(n, np.int32)
hit = np.zeros(n, bool)
for i in range(n):
    t, e = oracle.torus_first_hit(((0.0, 0.0, 0.0), R, r), o32[i], d32[i])
    ev[i], hit[i] = e, t is not None
# what setup() knows (FP64 restatement of TorusTest::setup, trt_device.hpp)
tc = -(o * d).sum(1)
q = o + tc[:, None] * d
m = (q * q).sum(1)
Rb2 = (R + r) ** 2 * (1 + 2.0 ** -9)
U = np.sqrt(np.maximum(Rb2 - m, 0))
lo, hi = np.maximum(0.001 - tc, -U), np.minimum(1e4 - tc, U)
rs = r * (1 + 2.0 ** -8)
with np.errstate(divide="ignore", invalid="ignore"):
    u0, u1 = (-rs - q[:, 1]) / d[:, 1], (rs - q[:, 1]) / d[:, 1]
lo, hi = np.maximum(lo, np.minimum(u0, u1)), np.minimum(hi, np.maximum(u0, u1))
kappa = m + R * R - r * r
a = d[:, 0] ** 2 + d[:, 2] ** 2
P2 = 2 * kappa - 4 * R * R * a
split = P2 < 0
w = np.sqrt(np.where(split, -P2 / 6, 0))
pieces = 1 + (split & (lo < -w) & (-w < hi)).astype(int) + (split & (lo < w) & (w < hi)).astype(int)
culled = ev == 0
c = (q[:, 0] ** 2 + q[:, 2] ** 2)
b = q[:, 0] * d[:, 0] + q[:, 2] * d[:, 2]
f_lo = lo ** 4 + P2 * lo ** 2 - 8 * R * R * b * lo + (kappa ** 2 - 4 * R * R * c)
fwd = (f_lo > 0)            # sign f(A) = sigma of the first piece (+1 outside the flex points): forward start
key = np.where(culled, 9, pieces * 2 + (~fwd).astype(int))
print(f"{n} aimed rays: {culled.mean() * 100:.1f} % culled in setup, {hit.mean() * 100:.1f} % hit, "
      f"{ev[~culled].mean():.2f} evaluations per solved test (max {ev.max()}); pieces 1/2/3: "
      + "/".join(f"{(pieces[~culled] == k).mean() * 100:.0f} %" for k in (1, 2, 3)))

F_CULL, F_FULL, RESETUP, SORT = 90, 270, 70, 30   # instructions per wave: culled-only wave / wave with survivors; extra work of a sort
waves = ev.reshape(-1, 64)
today_trips = waves.max(1).astype(float)
C_TRIP = (716 - F_FULL) / today_trips.mean()


def cost(order_ev, extra):
    """order_ev: walk lengths in the order the walk phase sees them (culled rays already dropped), padded to 64"""
    k = len(order_ev) // 64 * 64
    trips = order_ev[:k].reshape(-1, 64).max(1)
    return trips.sum() * C_TRIP, extra


def organisation(W, keyfn):
    total_walk = 0.0
    survivors = 0
    for s in range(0, n, W if W else n):
        e = ev[s:s + (W if W else n)]
        kk = keyfn(s, s + (W if W else n))
        idx = np.argsort(kk, kind="stable")
        e = e[idx]
        e = e[e > 0]
        survivors += len(e)
        pad = (-len(e)) % 64
        e = np.concatenate([e, np.zeros(pad, e.dtype)])
        total_walk += e.reshape(-1, 64).max(1).sum() * C_TRIP
    per64 = (total_walk + (n / 64) * (F_CULL + SORT) + (survivors / 64) * (F_FULL - F_CULL + RESETUP)) / (n / 64)
    return per64


base = 716.0
print(f"today: {base:.0f} wave-instructions per 64 rays (calibrated: {C_TRIP:.0f} per evaluation of the slowest lane, "
      f"{today_trips.mean():.2f} on average against {ev[~culled].mean():.2f} per solved test: lane utilisation of the walk "
      f"{ev.sum() / (today_trips.sum() * 64):.2f})")
for W in (256, 1024, 4096, 0):
    k = organisation(W, lambda s, e: key[s:e])
    i = organisation(W, lambda s, e: np.where(ev[s:e] == 0, 99, ev[s:e]))
    print(f"window {W or 'all':>5}: sorted by what setup() knows {k:5.0f} ({base / k:.2f}x)   sorted by the true walk length (bound) {i:5.0f} ({base / i:.2f}x)")
print("needed for 45 % of 8 TB/s from today's 31.5-32.5 %: 1.40-1.43x")
