#!/usr/bin/env python3
"""Per-step latency of the BiGRU recurrence (RMVPE, H=256): slope of wall time over T."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib
rng = np.random.default_rng(0)
H, I = 256, 384
sd = {}
for sfx in ("", "_reverse"):
    sd[f"fc.0.gru.weight_ih_l0{sfx}"] = (rng.standard_normal((3 * H, I)) * 0.05).astype(np.float32)
    sd[f"fc.0.gru.weight_hh_l0{sfx}"] = (rng.standard_normal((3 * H, H)) * 0.05).astype(np.float32)
    sd[f"fc.0.gru.bias_ih_l0{sfx}"] = np.zeros(3 * H, np.float32)
    sd[f"fc.0.gru.bias_hh_l0{sfx}"] = np.zeros(3 * H, np.float32)
ctx = _lib.Context(0)
ts = {}
for T in (1000, 3001, 9001):
    x = rng.standard_normal((1, T, I)).astype(np.float32)
    ctx.bigru(x, sd)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); ctx.bigru(x, sd); best = min(best, time.perf_counter() - t0)
    ts[T] = best
print("wall s:", ts, " us/step (slope 3001->9001): %.3f" % ((ts[9001] - ts[3001]) / 6000 * 1e6))
