#!/usr/bin/env python3
"""Per-launch time of the NSF upsamplers (ConvTranspose1d) through the conv profile (a HIP event pair around the launch).
usage: bench_convt.py   (A/B: RVCX_SHUF_STAGE=0|1, RVCX_CONVT_THIN=0|1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RVCX_DEBUG", "1")
import numpy as np
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib

SHAPES = [("48k stage 0", 512, 3199, 256, 24, 12, 6), ("48k stage 1", 256, 38377, 128, 20, 10, 5),
          ("40k stage 0", 512, 3199, 256, 16, 10, 3), ("40k stage 1", 256, 31991, 128, 16, 10, 3),
          ("48k stage 2", 128, 383761, 64, 4, 2, 1), ("48k stage 3", 64, 767521, 32, 4, 2, 1)]
ctx = _lib.Context(0)
g = np.random.Generator(np.random.PCG64(0))
for name, cin, tin, cout, k, s, p in SHAPES:
    x = g.standard_normal((1, cin, tin), dtype=np.float32)
    w = (g.standard_normal((cin, cout, k), dtype=np.float32) / (cin * k / s) ** 0.5).astype(np.float32)
    b = g.standard_normal(cout, dtype=np.float32)
    best = None
    for _ in range(4):
        ctx.conv_profile_begin()
        ctx.convtranspose1d(x, w, b, stride=s, pad=p, pre_lrelu=0.1)
        prof = ctx.conv_profile_end()
        ms = sum(r["ms"] for r in prof)
        best = ms if best is None else min(best, ms)
        tile = ", ".join(r["tile"].split(" (")[0] for r in prof)
    gf = 2.0 * cin * cout * k * tin / 1e9
    print(f"{name:12s} {cin:4d} -> {cout:4d} k {k:2d} s {s:2d} T {tin:7d}: {best * 1e3:8.1f} us  {gf / best:7.1f} TF/s  [{tile}]")
