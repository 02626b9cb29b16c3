import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu():
    try:
        import ctypes as C
        from bayesian_quadrature_amd import _lib
        n = C.c_int(0)
        _lib.load_library().bq_device_count(C.byref(n))
        return n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    from oracle import load
    o = load()
    o.set_threads(max(1, min(8, o.max_threads(), len(os.sched_getaffinity(0)))))
    return o


@pytest.fixture(scope="session")
def engine():
    if not _have_gpu():
        pytest.skip("no HIP device")
    from bayesian_quadrature_amd import get_engine
    return get_engine(0)


def rand_spd(rs, n):
    """Random SPD matrix as the reference's tests build it
    (tests/test_linalg_c.py:16-19): A + A^T + n I."""
    A = rs.rand(n, n)
    return np.asfortranarray(A + A.T + n * np.eye(n))
