#!/usr/bin/env python3
"""The two attention shapes of the path through rvcx_op_attention (run under rocprofv3 --kernel-trace --stats: the
kernel durations are what counts; the op itself includes host copies).  TextEncoder: T 3198, 2 heads x 96, window 10;
HuBERT: T 1599, 12 heads x 64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib
ctx = _lib.Context(0)
rng = np.random.default_rng(0)
for T, H, D, rel in ((3198, 2, 96, True), (1599, 12, 64, False)):
    q, k, v = (rng.standard_normal((1, H * D, T)).astype(np.float32) * 0.5 for _ in range(3))
    ek = (rng.standard_normal((1, 21, D)) * 0.1).astype(np.float32) if rel else None
    ev = (rng.standard_normal((1, 21, D)) * 0.1).astype(np.float32) if rel else None
    for _ in range(5):
        ctx.attention(q, k, v, H, D ** -0.5, ek, ev, 10)
print("done")
