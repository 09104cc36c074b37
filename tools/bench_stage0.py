#!/usr/bin/env python3
"""NSF stage-0 ResBlock convs (C = 256, T = 38376, k = 7 / 11): the two conv_h3 launches of a step with forced tiles.
usage: bench_stage0.py [tile indices into conv_h3's table, default 0,1,11,12]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib

tiles = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 11, 12]
ctx = _lib.Context(0)
for K, d in ((3, 1), (7, 3), (11, 5)):
    line = f"C=256 k={K:2d} d={d}: auto {sorted(ctx.bench_resblock_pair(1, 256, 38376, K, d, False, 5)[0] for _ in range(3))[1]:.3f} ms"
    for t in tiles:
        ctx.conv_override(100 + t, -1, -1)
        try:
            ms = sorted(ctx.bench_resblock_pair(1, 256, 38376, K, d, False, 5)[0] for _ in range(3))[1]
            line += f" | tile {t}: {ms:.3f}"
        except Exception as e:  # noqa
            line += f" | tile {t}: {str(e)[:30]}"
        ctx.conv_override(-1, -1, -1)
    print(line, flush=True)
