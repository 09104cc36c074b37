#!/usr/bin/env python3
"""Batched conversion vs single runs at full model size, bit for bit (debugging aid).  usage: [seconds] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib, synthetic as S, weights as W
ctx = _lib.Context(0)
seed = 1900
ctx.load_hubert(W.hubert_cfg_struct(S.HUBERT_CFG_BASE), S.hubert_state(S.HUBERT_CFG_BASE, seed))
ctx.load_rmvpe(W.rmvpe_cfg_struct(S.RMVPE_CFG_FULL), S.rmvpe_state(S.RMVPE_CFG_FULL, seed))
mid = ctx.load_synth(W.synth_cfg_struct(S.SYNTH_CFG_48K, 768), S.synth_state(S.SYNTH_CFG_48K, seed))
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
clips = [S.make_clip(20 + i, secs) for i in range(B)]
P = lambda s: _lib.Params(0.0, 50.0, 1100.0, 0.0, 0.33, 1.0, 0, 1, 6, 38, 41, s)
b1 = ctx.convert_batch(mid, clips, P(5), want_f32=True)[1]
b2 = ctx.convert_batch(mid, clips, P(5), want_f32=True)[1]
print("batch repeat equal:", [bool(np.array_equal(x, y)) for x, y in zip(b1, b2)])
for i, c in enumerate(clips):
    s1 = ctx.convert_batch(mid, [c], P(5 + i), want_f32=True)[1][0]
    s2 = ctx.convert_batch(mid, [c], P(5 + i), want_f32=True)[1][0]
    print(f"item {i}: single repeat equal {np.array_equal(s1, s2)}; batch == single {np.array_equal(b1[i], s1)} "
          f"(max diff {np.abs(b1[i] - s1).max():.2e}, first diff at {int(np.argmax(b1[i] != s1)) if not np.array_equal(b1[i], s1) else -1})")
print("fp32 reruns", ctx.fp32_reruns(), "gru fallbacks", ctx.gru_fallbacks())
