#!/usr/bin/env python3
"""conv_h3 tile / split-K sweep on the 3x3 convs of every RMVPE U-Net level, B = 1 and B = 8: what the launcher's
cost model picks against every forced (tile, split-K).  The 3x3 taps of a row-padded map are emulated by a 9-tap 1-D
conv whose dilation gives the same halo (2 Wp + 2), which is all the staging depends on."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RVCX_DEBUG", "1")   # tuning hooks are refused without it
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib
ctx = _lib.Context(0)
LEVELS = [("L0 C16", 16, 3232, 130), ("L1 C32", 32, 1616, 66), ("L2 C64", 64, 808, 34), ("L3 C128", 128, 404, 18),
          ("L4 C256", 256, 202, 10), ("L5 C512", 512, 101, 6)]
for B in (1, 8):
    for name, c, H, Wp in LEVELS:
        T, d = H * Wp, (2 * Wp + 2 + 7) // 8
        tiles = (103, 104, 109, 110) if 8 * d > 64 else (100, 101, 102)
        ctx.conv_override(-1, -1, -1)
        base, _ = ctx.bench_conv1d(B, c, T, c, 9, 1, d, 1, 10)
        res = []
        for t in tiles:
            for sk in (1, 2, 4, 8):
                ctx.conv_override(t, 0, sk)
                try:
                    ms, tf = ctx.bench_conv1d(B, c, T, c, 9, 1, d, 1, 10)
                    res.append((ms, t - 100, sk))
                except Exception:
                    pass
        res.sort()
        print(f"B={B} {name:8s} heuristic {base*1e3:7.1f} us | best " + " ".join(f"[t{t} s{s} {ms*1e3:.1f}]" for ms, t, s in res[:4]), flush=True)
ctx.conv_override(-1, -1, -1)
