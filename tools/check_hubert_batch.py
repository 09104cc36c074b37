#!/usr/bin/env python3
"""HuBERT features of a batch vs single runs, bit for bit, layer by layer (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib, synthetic as S, weights as W
ctx = _lib.Context(0)
cfg = S.HUBERT_CFG_BASE
ctx.load_hubert(W.hubert_cfg_struct(cfg), S.hubert_state(cfg, 1900))
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
a, b = S.make_clip(20, secs), S.make_clip(22, secs)
for L in (0, 1, 2, 12):
    s1 = ctx.hubert_features(a, 768, L)[0]
    s1b = ctx.hubert_features(a, 768, L)[0]
    s2 = ctx.hubert_features(b, 768, L)[0]
    bt = ctx.hubert_features(np.stack([a, b]), 768, L)
    print(f"layers {L:2d}: repeat equal {np.array_equal(s1, s1b)}; batch item0 equal {np.array_equal(bt[0], s1)} "
          f"(max diff {np.abs(bt[0] - s1).max():.2e}); item1 equal {np.array_equal(bt[1], s2)} (max diff {np.abs(bt[1] - s2).max():.2e})")
