import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import pytest

# the tuning / fault-injection hooks of librvcx.so (rvcx_conv_override, rvcx_debug_inject, rvcx_bench_*) are refused
# unless the process starts with RVCX_DEBUG=1; the kernel-level tests use them.  Read once at first use by the library.
os.environ.setdefault("RVCX_DEBUG", "1")


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
