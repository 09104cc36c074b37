#!/usr/bin/env python3
"""The same 30 s conversion N times: how many distinct results? (debugging aid)  usage: [seconds] [repeats]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib, synthetic as S, weights as W
ctx = _lib.Context(0)
seed = 1900
ctx.load_hubert(W.hubert_cfg_struct(S.HUBERT_CFG_BASE), S.hubert_state(S.HUBERT_CFG_BASE, seed))
ctx.load_rmvpe(W.rmvpe_cfg_struct(S.RMVPE_CFG_FULL), S.rmvpe_state(S.RMVPE_CFG_FULL, seed))
mid = ctx.load_synth(W.synth_cfg_struct(S.SYNTH_CFG_48K, 768), S.synth_state(S.SYNTH_CFG_48K, seed))
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
clip = S.make_clip(25, secs)
p = _lib.Params(0.0, 50.0, 1100.0, 0.0, 0.33, 1.0, 0, 1, 6, 38, 41, 5)
outs = [ctx.convert_batch(mid, [clip], p, want_f32=True)[1][0] for _ in range(n)]
hs = [hashlib.sha256(o.tobytes()).hexdigest()[:8] for o in outs]
print("distinct results:", len(set(hs)), hs, "max diff vs first", max(float(np.abs(o - outs[0]).max()) for o in outs))
