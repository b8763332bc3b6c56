#!/usr/bin/env python3
"""Makes tools/_aimed_ev.npz for tools/trace_sorted.py: bench.py's aimed ray set (2^20 rays, seed 1) with every ray's walk length
(polynomial evaluations of the CPU restatement's torus_first_hit; 0 = culled in setup).  CPU only, eight processes, a few seconds."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiprocessing import Pool
n = 1 << 20
rng = np.random.default_rng(1)
o = rng.uniform(-4, 4, (n, 3)); tgt = rng.normal(size=(n, 3))
tgt *= rng.uniform(0, 1.2, (n, 1)) / np.linalg.norm(tgt, axis=1, keepdims=True)
d = tgt - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
o32, d32 = o.astype(np.float32), d.astype(np.float32)
def work(rg):
    from oracle import oracle
    a, b = rg
    ev = np.zeros(b - a, np.int32)
    for i in range(a, b):
        t, e = oracle.torus_first_hit(((0.0, 0.0, 0.0), 1.0, 0.25), o32[i], d32[i])
        ev[i - a] = e
    return ev
if __name__ == "__main__":
    parts = [(k * n // 8, (k + 1) * n // 8) for k in range(8)]
    with Pool(8) as p: ev = np.concatenate(p.map(work, parts))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), '_aimed_ev.npz'), o=o32, d=d32, ev=ev)
    print(n, ev.mean(), ev.max(), (ev == 0).mean())
