#!/usr/bin/env python3
"""A/B of the F0 model between two builds (RVCX_LIBRARY): sha1 of f0 over a 30 s clip (full-size RMVPE, synthetic weights)
and of a ragged batch of three clips; time per call.  usage: ab_f0.py"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib, synthetic as S, weights as W

if __name__ == "__main__":
    ctx = _lib.Context(0)
    cfg = S.RMVPE_CFG_FULL
    ctx.load_rmvpe(W.rmvpe_cfg_struct(cfg), S.rmvpe_state(cfg, 7))
    x = S.make_clip(5, 32.0).astype(np.float32)
    f0 = ctx.rmvpe_f0(x)
    for _ in range(3):
        ctx.rmvpe_f0(x)
    t0 = time.perf_counter()
    for _ in range(20):
        ctx.rmvpe_f0(x)
    dt = (time.perf_counter() - t0) / 20
    print(f"{os.environ.get('RVCX_LIBRARY', 'in-tree')}: f0 sha1 {hashlib.sha1(f0.tobytes()).hexdigest()[:16]} voiced {np.mean(f0 > 0):.3f} "
          f"{dt * 1e3:.3f} ms per call")
