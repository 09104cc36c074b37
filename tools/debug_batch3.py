#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib, synthetic as S, weights as W
ctx = _lib.Context(0)
ctx.load_rmvpe(W.rmvpe_cfg_struct(S.RMVPE_CFG_FULL), S.rmvpe_state(S.RMVPE_CFG_FULL, 1900))
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 32.0
wav = np.stack([S.make_clip(i, secs) for i in range(8)])
for B in (1, 2, 3, 4, 8):
    f0, hid = ctx.rmvpe_f0(wav[:B], return_hidden=True)
    print(f"{os.environ.get('TAG','')} rmvpe {secs}s B={B}: finite per item", [bool(np.isfinite(hid[b]).all()) for b in range(B)], flush=True)
    if B == 1: h1 = hid[0]
    else: print("   item0 equal:", np.array_equal(h1, hid[0]), flush=True)
