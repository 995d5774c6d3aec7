import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def engine():
    """One pgx context for the whole GPU session (fails loudly if libpgx.so / the GPU is missing)."""
    import photogrammetry_amd as pg
    e = pg.Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="session")
def lego():
    return dict(np.load(os.path.join(GOLDEN, "lego_descriptors.npz")))


@pytest.fixture(scope="session")
def star():
    return dict(np.load(os.path.join(GOLDEN, "star_pair.npz")))


def star_rgba64(star, tag):
    """Rebuild the RGBA64 image the way ImageSharp widens an 8-bit PNG (x257)."""
    H, W = 383, 451
    rgb = np.unpackbits(star[tag + "_bits"])[:H * W * 3].reshape(H, W, 3).astype(np.uint16) * 65535
    a = np.unpackbits(star[tag + "_alpha"])[:H * W].reshape(H, W).astype(np.uint16) * 65535
    return np.ascontiguousarray(np.concatenate([rgb, a[..., None]], axis=2))


def pairs_arr(p):
    return np.stack([p["k1"], p["k2"], p["dist"]], axis=1).astype(np.int64)
