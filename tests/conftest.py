import os
import sys

import numpy as np
import pytest

try:  # torch first: a process holds one HIP runtime, and torch must find the one it was built with
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def gctx():
    """One orbamd context on GPU 0 for the whole GPU test session."""
    from visual_slam_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def make_test_image(H, W, seed, nrect=None, noise=6):
    """Structured gray image with many corners at all orientations: random filled rectangles and
    single-pixel / 2x2 specks over a noisy mid-gray background."""
    rng = np.random.default_rng(seed)
    img = np.full((H, W), 110, np.int32)
    if nrect is None:
        nrect = max(4, (H * W) // 900)
    for _ in range(nrect):
        h, w = int(rng.integers(2, 24)), int(rng.integers(2, 24))
        y, x = int(rng.integers(-4, H)), int(rng.integers(-4, W))
        img[max(y, 0):max(y + h, 0), max(x, 0):max(x + w, 0)] = int(rng.integers(0, 256))
    for _ in range(nrect):
        y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
        s = int(rng.integers(1, 3))
        img[y:y + s, x:x + s] = int(rng.integers(0, 256))
    img = img + rng.integers(-noise, noise + 1, size=(H, W))
    return np.clip(img, 0, 255).astype(np.uint8)
