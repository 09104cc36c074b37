import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import pytest


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    import polgen_rvc_amd  # noqa: F401
    from polgen_rvc_amd import _lib
    c = _lib.Context(0)
    yield c
    c.close()


def rms(a):
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(np.mean(a * a)))
