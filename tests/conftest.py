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


# Parity bars of the FULL-SIZE conversions (C1 / C2 / C3 / C5 against the reference's fixtures).  The north star allows
# 1e-3 RMS on the float waveform; what is measured is 0.8e-5 .. 1.0e-5 (signal RMS 0.18) and 1-2 LSB on the strided PCM
# samples, so the bars sit at 3x that: a 5x numerical regression fails (VERDICT r4 "tighten to what is measured").
FULL_RMS_BAR = 3e-5
FULL_PCM_BAR = 4
TINY_RMS_BAR = 2e-5      # tiny-config goldens: measured 2e-6 .. 4e-6


def rms(a):
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(np.mean(a * a)))
