import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))  # tests may use the oracle as the checker


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    import __graft_entry__ as g
    g.build()


@pytest.fixture(scope="session")
def small_scene():
    from fspt_amd import scene as S
    return S.bunny_scene(n=8, env_size=(64, 32))


@pytest.fixture(scope="session")
def medium_scene():
    from fspt_amd import scene as S
    return S.bunny_scene(n=24, env_size=(256, 128), sun_deg=3.0)


@pytest.fixture(scope="session")
def camera():
    from fspt_amd import scene as S
    c = dict(S.BUNNY_CAMERA)
    c["lens"] = S.lens_features(c["focal_depth"], c["aperture"])
    return c


def random_rays(arrays, n, seed=0):
    """Rays aimed at the scene from a shell around it (plus some grazing/away rays)."""
    rng = np.random.default_rng(seed)
    tri = arrays.tri.reshape(-1, 3, 3)
    lo, hi = tri.reshape(-1, 3).min(0), tri.reshape(-1, 3).max(0)
    c, r = (lo + hi) / 2, np.linalg.norm(hi - lo) / 2
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = c + d * r * rng.uniform(0.2, 2.0, size=(n, 1))
    tgt = c + rng.normal(size=(n, 3)) * r * 0.6
    dirs = tgt - o
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    return np.concatenate([o, dirs], 1).astype(np.float32)
