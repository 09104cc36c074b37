#!/usr/bin/env python3
"""1-D conv shapes of the decoder / front end under each halo-64 tile of conv_h3 (run once per tile:
RVCX_CONV_TILE=100 (32 x 256), 101 (64 x 128), 102 (64 x 64), unset = the cost model's choice)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RVCX_DEBUG", "1")   # tuning hooks are refused without it
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib
ctx = _lib.Context(0)
SH = [("ups3-like 64->64 k2", 1, 64, 767521, 64, 2, 1), ("ups2-like 128->128 k2", 1, 128, 383761, 128, 2, 1),
      ("ups1-like 256->1280 k2", 1, 256, 38377, 1280, 2, 1), ("ups0-like 512->3072 k2", 1, 512, 3199, 3072, 2, 1),
      ("conv_pre 192->512 k7", 1, 192, 3198, 512, 7, 1), ("s0 256->256 k7", 1, 256, 38376, 256, 7, 1),
      ("s0 256->256 k11 d5", 1, 256, 38376, 256, 11, 5), ("s0 256->256 k3", 1, 256, 38376, 256, 3, 1),
      ("128->128 k7", 1, 128, 383760, 128, 7, 1), ("64->64 k3", 1, 64, 767520, 64, 3, 1), ("32->32 k3", 1, 32, 1535040, 32, 3, 1),
      ("flow in 192->384 k5", 1, 192, 3198, 384, 5, 1), ("ffn1 192->768 k3", 1, 192, 3198, 768, 3, 1),
      ("ffn2 768->192 k3", 1, 768, 3198, 192, 3, 1), ("B16 flow in", 16, 192, 3198, 384, 5, 1), ("B16 s0 k7", 16, 256, 38376, 256, 7, 1),
      ("B16 ups1-like", 16, 256, 38377, 1280, 2, 1)]
for name, B, Cin, Tin, Cout, K, d in SH:
    ms, tf = ctx.bench_conv1d(B, Cin, Tin, Cout, K, 1, d, 1, 4)
    print(f"{name:26s} {ms*1e3:9.1f} us {tf:7.1f} TF/s")
