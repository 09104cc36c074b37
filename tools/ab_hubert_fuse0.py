#!/usr/bin/env python3
"""HuBERT features of a 32 s clip (HuBERT-base, synthetic weights) under the current environment: saves them to argv[1] and
prints the time per call; run twice (RVCX_HUBERT_FUSE0=0 / unset) and compare with `ab_hubert_fuse0.py a.npy b.npy`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if len(sys.argv) == 3:
    a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
    d = a.astype(np.float64) - b
    print(f"features {a.shape}: max |diff| {np.abs(d).max():.3e}, rel rms {np.sqrt((d * d).mean() / (a.astype(np.float64) ** 2).mean()):.3e}")
    sys.exit(0)
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib, synthetic as S, weights as W
ctx = _lib.Context(0)
cfg = S.HUBERT_CFG_BASE
ctx.load_hubert(W.hubert_cfg_struct(cfg), S.hubert_state(cfg, 7))
x = S.make_clip(5, 32.0).astype(np.float32)[None]
f = ctx.hubert_features(x, cfg["embed_dim"], cfg["layers"])
for B in (1, 8):
    xb = np.repeat(x, B, axis=0)
    for _ in range(2):
        ctx.hubert_features(xb, cfg["embed_dim"], cfg["layers"])
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.hubert_features(xb, cfg["embed_dim"], cfg["layers"])
    print(f"FUSE0={os.environ.get('RVCX_HUBERT_FUSE0', '1')} B={B}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per call (host copies included)")
np.save(sys.argv[1], f)
