#!/usr/bin/env python3
"""h3 tile sweep on the NSF / HuBERT / enc_p shapes (forced tiles 100+t, split-K s)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib
SH = [("nsf1 C256 k3", 256, 38376, 256, 3, 1, 1), ("nsf1 C256 k11", 256, 38376, 256, 11, 1, 1),
      ("nsf2 C128 k3", 128, 383760, 128, 3, 1, 1), ("nsf2 C128 k7", 128, 383760, 128, 7, 1, 1), ("nsf2 C128 k11", 128, 383760, 128, 11, 1, 1),
      ("nsf3 C64 k3", 64, 767520, 64, 3, 1, 1), ("nsf3 C64 k7", 64, 767520, 64, 7, 1, 1), ("nsf3 C64 k11", 64, 767520, 64, 11, 1, 1),
      ("nsf4 C32 k3", 32, 1535040, 32, 3, 1, 1), ("nsf4 C32 k7", 32, 1535040, 32, 7, 1, 1), ("nsf4 C32 k11", 32, 1535040, 32, 11, 1, 1),
      ("hubert conv1 s2", 512, 102399, 512, 3, 2, 1), ("hubert conv3 s2", 512, 25599, 512, 3, 2, 1),
      ("hubert qkv", 768, 1599, 2304, 1, 1, 1), ("hubert o", 768, 1599, 768, 1, 1, 1), ("hubert fc1", 768, 1599, 3072, 1, 1, 1),
      ("hubert fc2", 3072, 1599, 768, 1, 1, 1), ("enc_p ffn 192-768 k3", 192, 3198, 768, 3, 1, 1), ("enc_p ffn 768-192 k3", 768, 3198, 192, 3, 1, 1),
      ("flow wn k5", 192, 3198, 384, 5, 1, 1)]
ctx = _lib.Context(0)
it = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for name, cin, T, cout, k, st, d in SH:
    ctx.conv_override(-1, -1, -1)
    base, _ = ctx.bench_conv1d(1, cin, T, cout, k, st, d, 1, it)
    res = []
    for t in range(9):
        for sk in (1, 2, 4, 8):
            if sk > 1 and T > 8000:
                continue
            ctx.conv_override(100 + t, 0, sk)
            ms, tf = ctx.bench_conv1d(1, cin, T, cout, k, st, d, 1, it)
            res.append((ms, t, sk, tf))
    res.sort()
    seen, top = set(), []
    for ms, t, sk, tf in res:       # forced tiles that do not apply fall back to the heuristic: keep distinct times
        if (round(ms, 4)) in seen:
            continue
        seen.add(round(ms, 4)); top.append(f"[h{t} s{sk} {ms:.3f}ms {tf:.0f}TF]")
    print(f"{name:24s} heuristic {base:.3f} ms | " + " ".join(top[:5]), flush=True)
