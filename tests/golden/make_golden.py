"""Regenerates the golden fixtures in tests/golden/ (run from the repo root:
``python tests/golden/make_golden.py``).

The reference holds NO golden vectors, tests or fixtures for this path,
and cannot be run, so nothing here comes from the reference.  The fixtures are:

* ``kat.json``        analytic known-answer vectors k1-k8 of SURVEY.md §8c (typed in, exact
                      rational arithmetic done by hand / verified with numpy.roots there);
* ``rays_*.npz``      seeded random rays with the first-hit parameter computed by the
                      independent FP64 companion-matrix solver (oracle/truth.py) and a
                      "robust" flag (classification insensitive to a 1e-4 perturbation);
* ``render_*.npz``    small frames rendered by the CPU oracle (regression pins for the
                      oracle itself and a second comparison target for the GPU tests).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import seeded_rays  # noqa: E402
from oracle import oracle, truth  # noqa: E402
from toroidal_ray_tracing_amd import abi, camera  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def kat():
    s = 1.0 / np.sqrt(1 + 0.02 ** 2 + 0.05 ** 2)
    return {
        "torus": {"center": [0, 0, 0], "R": 1.0, "r": 0.25},
        "rays": [
            {"name": "k1", "o": [-5, 0, 0], "d": [1, 0, 0], "roots": [3.75, 4.25, 5.75, 6.25],
             "t": 3.75, "P": [-1.25, 0, 0], "N": [-1, 0, 0]},
            {"name": "k2", "o": [1, 5, 0], "d": [0, -1, 0], "roots": [4.75, 5.25],
             "t": 4.75, "P": [1, 0.25, 0], "N": [0, 1, 0]},
            {"name": "k3", "o": [0, 5, 0], "d": [0, -1, 0], "roots": [], "t": None},
            {"name": "k4", "o": [-3, 0.1, 0.2], "d": [s, 0.02 * s, -0.05 * s],
             "roots": [1.797800640, 2.212485692, 3.828395781, 4.175909342], "t": 1.797800640},
            {"name": "k5", "o": [-1.25, 0, 0], "d": [1, 0, 0], "roots": [0.0, 0.5, 2.0, 2.5],
             "t": 0.5, "P": [-0.75, 0, 0], "N": [1, 0, 0]},
            {"name": "k6", "o": [-5, 2, 0], "d": [1, 0, 0], "roots": [], "t": None},
        ],
        "reflect": {"name": "k7", "I": [1, 0, 0], "N": [-1, 0, 0], "R": [-1, 0, 0]},
        "tmin": 0.001, "tmax": 10000.0,
    }


def ray_fixture(name, tori, n, seed, center=(0, 0, 0)):
    o, d = seeded_rays(n, seed, center=center)
    t, tid = truth.first_hit(o, d, tori)
    robust = truth.classify_margin(o, d, tori)
    np.savez_compressed(os.path.join(HERE, name), o=o, d=d, t=t, id=tid.astype(np.int32), robust=robust,
                        tori=np.array([[*c, R, r] for c, R, r in tori], np.float64))


def render_fixture(name, scene, g, pc, W, H, cam, precision=abi.TRT_SOLVE_F32):
    rgba, hits, rendered, stats = oracle.render(scene, g, pc, W, H, cam, precision=precision,
                                                want_rendered=True)
    np.savez_compressed(os.path.join(HERE, name), rgba=rgba, rendered=rendered,
                        stats=np.array([stats[k] for k in ("primary_tests", "bounce_tests", "shadow_tests")]),
                        **{"hit_" + k: v for k, v in hits.items()})


def main():
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat(), f, indent=1)
    ray_fixture("rays_single.npz", [((0, 0, 0), 1.0, 0.25)], 4096, 0x5EED)
    ray_fixture("rays_thin_offset.npz", [((0.3, -0.2, 0.5), 2.0, 0.1)], 2048, 0x5EED + 1, center=(0.3, -0.2, 0.5))
    ray_fixture("rays_nested.npz", [((0, 0, 0), 1.0, 0.05 * (i + 1)) for i in range(8)], 2048, 0x5EED + 2)
    W = H = 64
    render_fixture("render_pinhole_mirror.npz", camera.single_torus_scene(), camera.baseline_camera(W, H),
                   camera.baseline_push(5), W, H, abi.TRT_CAMERA_PINHOLE)
    render_fixture("render_pinhole_nested_f64.npz", camera.nested_tori_scene(), camera.baseline_camera(W, H),
                   camera.baseline_push(5), W, H, abi.TRT_CAMERA_PINHOLE, abi.TRT_SOLVE_F64)
    pc = camera.baseline_push(5)
    pc.rho = 4.0
    # toroidal camera inside a torus ("tokamak interior"): origins on the rho=4 circle around the
    # eye, inside the hole of a R=6, r=1.5 torus; eye/center as in BEF/main.cpp:124
    render_fixture("render_toroidal_plastic.npz",
                   camera.single_torus_scene(center=(0.0, 0.0, 0.0), R=6.0, r=1.5, material=camera.PLASTIC),
                   camera.toroidal_camera(W, H), pc, W, H, abi.TRT_CAMERA_TOROIDAL)


if __name__ == "__main__":
    main()
