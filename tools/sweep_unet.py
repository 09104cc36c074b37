#!/usr/bin/env python3
"""h3 tile / split-K sweep on the deep RMVPE U-Net levels (3x3 convs whose 2*Wp+2 halo fits the 1-D tiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib
ctx = _lib.Context(0)
for name, c, T in [("L2 C64", 64, 27472), ("L3 C128", 128, 7272), ("L4 C256", 256, 2020), ("L5 C512", 512, 606)]:
    ctx.conv_override(-1, -1, -1)
    base, _ = ctx.bench_conv1d(1, c, T, c, 9, 1, 4, 1, 20)
    res = []
    for t in (100, 101, 102):
        for sk in (1, 2, 4, 8):
            ctx.conv_override(t, 0, sk)
            ms, tf = ctx.bench_conv1d(1, c, T, c, 9, 1, 4, 1, 20)
            res.append((ms, t - 100, sk))
    res.sort()
    print(f"{name:8s} heuristic {base*1e3:.1f} us | " + " ".join(f"[h{t} s{s} {ms*1e3:.1f}us]" for ms, t, s in res[:5]), flush=True)
