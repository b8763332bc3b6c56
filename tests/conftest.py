import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def seeded_rays(n, seed, aimed_fraction=0.7, box=4.0, reach=1.6, center=(0.0, 0.0, 0.0)):
    """Random rays: origins uniform in [-box,box]^3 (+center); a fraction aimed at a ball of
    radius `reach` around `center` (so that many hit), the rest with directions uniform on S²."""
    rng = np.random.default_rng(seed)
    c = np.asarray(center, np.float64)
    o = rng.uniform(-box, box, (n, 3)) + c
    d = rng.normal(size=(n, 3))
    tgt = rng.normal(size=(n, 3))
    tgt *= (rng.uniform(0, reach, (n, 1)) / np.linalg.norm(tgt, axis=1, keepdims=True))
    aimed = rng.uniform(size=n) < aimed_fraction
    d[aimed] = (tgt + c - o)[aimed]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o.astype(np.float32), d.astype(np.float32)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.lib()
    return o
