#!/usr/bin/env python3
"""Per-launch table of HuBERT alone: B clips of `seconds` s (+ 2 s of padding) through rvcx_hubert_features with the
conv / GEMM profile on (a HIP event pair around every launch, one stream).  usage: prof_hubert.py [B=1] [seconds=30]"""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib, synthetic as S, weights as W

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sec = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
ctx = _lib.Context(0)
ctx.load_hubert(W.hubert_cfg_struct(S.HUBERT_CFG_BASE), S.hubert_state(S.HUBERT_CFG_BASE, 1900))
wav = np.stack([np.pad(S.make_clip(25 + b, sec), (16000, 16000), mode="reflect").astype(np.float32) for b in range(B)])
for _ in range(3):
    ctx.hubert_features(wav, 768, 12)
ts = []
for _ in range(6):
    t0 = time.perf_counter()
    ctx.hubert_features(wav, 768, 12)
    ts.append(time.perf_counter() - t0)
print(f"B={B} {sec:.0f} s: HuBERT wall best {min(ts)*1e3:.2f} ms ({min(ts)*1e3/B:.2f} per clip; includes H2D / D2H of the op)")
ctx.conv_profile_begin()
ctx.hubert_features(wav, 768, 12)
prof = ctx.conv_profile_end()
import csv
rows = list(csv.reader(ctx.conv_profile_csv().strip().splitlines()[1:]))    # tile names hold commas: quoted
acc = collections.OrderedDict()
for r in rows:
    key = (r[2], r[3], r[4], r[5], r[6], r[0])
    a = acc.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += float(r[8])
    a[2] += float(r[7])
tot = sum(a[1] for a in acc.values())
print(f"conv / GEMM launches {len(rows)}, ms {tot:.3f}")
print(f"{'cin':>5s} {'cout':>5s} {'k':>3s} {'st':>3s} {'nout':>8s} {'tile':>4s} {'n':>4s} {'ms':>8s} {'us/launch':>10s} {'TF/s':>7s}")
for (cin, cout, k, st, nout, tile), (n, ms, gf) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{cin:>5s} {cout:>5s} {k:>3s} {st:>3s} {nout:>8s} {tile:>4s} {n:4d} {ms:8.3f} {ms/n*1e3:10.1f} {gf/ms:7.1f}")
