#!/usr/bin/env python3
"""Tile / staging-variant / split-K sweep over the stride-1 conv shapes of the C2 workload (GPU box).
Prints, per shape, the heuristic's time and the best forced configuration: input for tuning
launch_conv_fast() in csrc/conv_fast.hip.  usage: sweep_conv.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib

# (name, Cin, T, Cout, K, dil)
SHAPES = [
    ("nsf1 C256 k3", 256, 38376, 256, 3, 1), ("nsf1 C256 k7", 256, 38376, 256, 7, 1),
    ("nsf1 C256 k11", 256, 38376, 256, 11, 1), ("nsf1 C256 k11d5", 256, 38376, 256, 11, 5),
    ("nsf2 C128 k3", 128, 383760, 128, 3, 1), ("nsf2 C128 k7", 128, 383760, 128, 7, 1),
    ("nsf2 C128 k11", 128, 383760, 128, 11, 1),
    ("nsf3 C64 k3", 64, 767520, 64, 3, 1), ("nsf3 C64 k7", 64, 767520, 64, 7, 1),
    ("nsf3 C64 k11", 64, 767520, 64, 11, 1),
    ("nsf4 C32 k3", 32, 1535040, 32, 3, 1), ("nsf4 C32 k7", 32, 1535040, 32, 7, 1),
    ("nsf4 C32 k11", 32, 1535040, 32, 11, 1),
    ("hubert qkv", 768, 1599, 2304, 1, 1), ("hubert o", 768, 1599, 768, 1, 1),
    ("hubert fc1", 768, 1599, 3072, 1, 1), ("hubert fc2", 3072, 1599, 768, 1, 1),
    ("enc_p ffn k3 192-768", 192, 3198, 768, 3, 1), ("enc_p ffn k3 768-192", 768, 3198, 192, 3, 1),
    ("enc_p qkv", 192, 3198, 576, 1, 1), ("enc_p o", 192, 3198, 192, 1, 1),
    ("flow wn k5", 192, 3198, 384, 5, 1), ("flow rs", 192, 3198, 384, 1, 1),
    ("rmvpe-ish C512 T606 k9", 512, 606, 512, 9, 1),
]
TILES = {False: range(0, 6), True: range(9, 12)}
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = _lib.Context(0)
ALL = []
for name, cin, T, cout, k, dil in SHAPES:
    lin = k == 1 and cin % 32 == 0
    ctx.conv_override(-1, -1, -1)
    base, _ = ctx.bench_conv1d(1, cin, T, cout, k, 1, dil, 1, iters)
    res = []
    for tile in TILES[lin]:
        for variant in (0,):
            for sk in (1, 2, 4, 8):
                if sk > 1 and T > 8000:
                    continue
                ctx.conv_override(tile, variant, sk)
                try:
                    ms, tf = ctx.bench_conv1d(1, cin, T, cout, k, 1, dil, 1, iters)
                except Exception as e:  # noqa
                    continue
                res.append((ms, tile, variant, sk, tf))
    res.sort()
    ALL.append(dict(name=name, cin=cin, T=T, cout=cout, k=k, dil=dil, heuristic_ms=base,
                    results=[dict(ms=ms, tile=t, variant=v, splitk=sk) for ms, t, v, sk, tf in res]))
    top = "  ".join(f"[t{t} v{v} s{s} {ms:.3f}ms {tf:.0f}TF]" for ms, t, v, s, tf in res[:4])
    print(f"{name:26s} heuristic {base:.3f} ms | {top}", flush=True)
import json
os.makedirs("gpurun_out", exist_ok=True)
json.dump(ALL, open("gpurun_out/sweep_conv.json", "w"))
