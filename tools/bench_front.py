#!/usr/bin/env python3
"""HuBERT and the F0 model ALONE on a 30 s clip (wall time of the op incl. H2D of the padded audio and D2H of the
result; best of N): what each front-end branch costs without the other beside it.  usage: bench_front.py [reps=10]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib, synthetic as S, weights as W

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = _lib.Context(0)
ctx.load_hubert(W.hubert_cfg_struct(S.HUBERT_CFG_BASE), S.hubert_state(S.HUBERT_CFG_BASE, 1900))
ctx.load_rmvpe(W.rmvpe_cfg_struct(S.RMVPE_CFG_FULL), S.rmvpe_state(S.RMVPE_CFG_FULL, 1900))
wav = np.pad(S.make_clip(25, 30.0), (16000, 16000), mode="reflect").astype(np.float32)   # x_pad = 1 s each side


def best(f):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3


h = best(lambda: ctx.hubert_features(wav, 768, 12))
r = best(lambda: ctx.rmvpe_f0(wav))
print(f"HuBERT alone: best {h[0]:.2f} ms, median {h[1]:.2f} ms   F0 model (mel + U-Net + BiGRU + decode) alone: best {r[0]:.2f} ms, median {r[1]:.2f} ms")
