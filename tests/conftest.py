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


def _gpu_count():
    try:
        from tomo_tv_amd import _lib
        return _lib.device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu():
    """GPU tests must run the HIP path: fail (not skip) when selected with -m gpu but no device/library."""
    n = _gpu_count()
    if n == 0:
        pytest.fail("no HIP device or libtomo_hip.so missing: the product has no CPU fallback")
    return n


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    d = np.linalg.norm(b.ravel())
    return float(np.linalg.norm((a - b).ravel()) / (d if d > 0 else 1.0))
