#!/usr/bin/env python3
"""A/B of the high-pass filter (filtfilt, csrc/pipeline.hip) between two builds of the library: run once per build with
RVCX_LIBRARY pointing at it; prints a checksum of the float64 output and the time of 50 back-to-back calls on a 30 s clip.
usage: ab_highpass.py [seconds=30]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib, synthetic as S

if __name__ == "__main__":
    sec = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    ctx = _lib.Context(0)
    x = S.make_clip(5, sec).astype(np.float64)
    y = ctx.highpass(x)
    for _ in range(3):
        ctx.highpass(x)
    t0 = time.perf_counter()
    for _ in range(50):
        ctx.highpass(x)
    dt = (time.perf_counter() - t0) / 50
    print(f"{os.environ.get('RVCX_LIBRARY', 'in-tree')}: sha1 {hashlib.sha1(y.tobytes()).hexdigest()[:16]}  "
          f"{dt * 1e3:.3f} ms per call (host copies included), |y| max {np.abs(y).max():.4f}")
