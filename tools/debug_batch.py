#!/usr/bin/env python3
"""Debug helper: batched conversion at full size, finite-ness and equality per item."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib, synthetic as S, weights as W

SEED = 1900
ctx = _lib.Context(0)
ctx.load_hubert(W.hubert_cfg_struct(S.HUBERT_CFG_BASE), S.hubert_state(S.HUBERT_CFG_BASE, SEED))
ctx.load_rmvpe(W.rmvpe_cfg_struct(S.RMVPE_CFG_FULL), S.rmvpe_state(S.RMVPE_CFG_FULL, SEED))
mid = ctx.load_synth(W.synth_cfg_struct(S.SYNTH_CFG_48K, 768), S.synth_state(S.SYNTH_CFG_48K, SEED))


def P(index_rate=0.0, seed=5):
    return _lib.Params(0.0, 50.0, 1100.0, index_rate, 0.33, 1.0, 0, 1, 6, 38, 41, seed)


secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
clips = [S.make_clip(i, secs) for i in range(8)]
for B in (2, 8):
    pcm, f32 = ctx.convert_batch(mid, clips[:B], P(), want_f32=True)
    print(f"no index B={B}: finite per item", [bool(np.isfinite(x).all()) for x in f32], flush=True)
big = S.make_index(65536, 768, 0)
ctx.load_index(big)
for B in (1, 2, 8):
    pcm, f32 = ctx.convert_batch(mid, clips[:B], P(0.75), want_f32=True)
    print(f"index B={B}: finite per item", [bool(np.isfinite(x).all()) for x in f32], flush=True)
    if B == 1:
        ref0 = f32[0]
    else:
        print("   item0 equals single:", np.array_equal(ref0, f32[0]), flush=True)
