#!/usr/bin/env python3
"""A/B micro-benchmark of one ResBlock1 step on the NSF decoder shapes of the 30 s / 48 k workload: the fused
kernel (resblock.hip) against the two conv_h3 launches it replaces, alternating in one process (the GPU's power
state drifts over seconds: never compare numbers from different processes).
usage: bench_pair.py [rounds=3] [only_fused=0] [shape indices, e.g. 2,5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RVCX_DEBUG", "1")   # tuning hooks are refused without it
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib

SHAPES = [(256, 38376, 3, 1), (128, 383760, 3, 1), (128, 383760, 7, 3), (128, 383760, 11, 5), (64, 767520, 3, 1), (64, 767520, 7, 3),
          (64, 767520, 11, 5), (32, 1535040, 3, 1), (32, 1535040, 7, 3), (32, 1535040, 11, 5)]
if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    only_fused = len(sys.argv) > 2 and sys.argv[2] == "1"
    if len(sys.argv) > 3:
        SHAPES = [SHAPES[int(v)] for v in sys.argv[3].split(",")]
    ctx = _lib.Context(0)
    tot = {True: 0.0, False: 0.0}
    for C, T, K, d in SHAPES:
        res = {True: [], False: []}
        for _ in range(rounds):
            for fused in ((True,) if only_fused else (True, False)):
                res[fused].append(ctx.bench_resblock_pair(1, C, T, K, d, fused, 5))
        line = f"C={C:3d} k={K:2d} d={d}: "
        for fused in ((True,) if only_fused else (True, False)):
            ms = sorted(r[0] for r in res[fused])[len(res[fused]) // 2]
            tf = 4.0 * C * C * K * T / (ms * 1e-3) / 1e12
            tot[fused] += ms
            line += f"{'fused' if fused else 'two  '} {ms:7.3f} ms {tf:6.1f} TF/s ({tf / 833.3 * 100:4.1f} %)   "
        print(line, flush=True)
    print("sum over the 9 (C, k) pairs:", {("fused" if k else "two"): round(v, 3) for k, v in tot.items()})
