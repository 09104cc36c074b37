#!/usr/bin/env python3
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
for secs, B in [(5.0, 8), (15.0, 8), (30.0, 4), (30.0, 8)]:
    clips = [S.make_clip(i, secs) for i in range(B)]
    pcm, f32 = ctx.convert_batch(mid, clips, P(), want_f32=True)
    print(f"convert {secs}s B={B}: finite", [bool(np.isfinite(x).all()) for x in f32], flush=True)
n = 512000
wav = np.stack([S.make_clip(i, 32.0) for i in range(8)])
for B in (1, 4, 8):
    f0, hid = ctx.rmvpe_f0(wav[:B], return_hidden=True)
    print(f"rmvpe B={B}: finite {np.isfinite(hid).all()} f0 {np.isfinite(f0).all()}", flush=True)
    if B == 1: h1 = hid[0]
    else: print("   item0 equal:", np.array_equal(h1, hid[0]), flush=True)
for B in (1, 4, 8):
    ft = ctx.hubert_features(wav[:B], 768)
    print(f"hubert B={B}: finite {np.isfinite(ft).all()}", [bool(np.isfinite(ft[b]).all()) for b in range(B)], flush=True)
    if B == 1: f1 = ft[0]
    else: print("   item0 equal:", np.array_equal(f1, ft[0]), flush=True)
T = 3198
g = np.random.Generator(np.random.PCG64(0))
phone = g.standard_normal((8, T, 768)).astype(np.float32)
pitch = g.integers(1, 255, (8, T)).astype(np.int32)
pitchf = (100 + 200 * g.random((8, T))).astype(np.float32)
for B in (1, 4, 8):
    out = ctx.synth_infer(mid, phone[:B], pitch[:B], pitchf[:B], seed=3)
    print(f"synth B={B}: finite", [bool(np.isfinite(out[b]).all()) for b in range(B)], flush=True)
    if B == 1: o1 = out[0]
    else: print("   item0 equal:", np.array_equal(o1, out[0]), flush=True)
