#!/usr/bin/env python3
"""Micro-benchmark of the MFMA conv kernel on the layer shapes of the 30 s / 48 k workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RVCX_DEBUG", "1")   # tuning hooks are refused without it
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib

SHAPES = [
    # name, B, Cin, Tin, Cout, K, stride, dil
    ("nsf s1 C256 k3", 1, 256, 38376, 256, 3, 1, 1),
    ("nsf s1 C256 k11", 1, 256, 38376, 256, 11, 1, 5),
    ("nsf s2 C128 k7", 1, 128, 383760, 128, 7, 1, 3),
    ("nsf s3 C64 k7", 1, 64, 767520, 64, 7, 1, 1),
    ("nsf s4 C32 k3", 1, 32, 1535040, 32, 3, 1, 1),
    ("nsf s4 C32 k11", 1, 32, 1535040, 32, 11, 1, 5),
    ("hubert conv1 k3s2", 1, 512, 102399, 512, 3, 2, 1),
    ("hubert qkv", 1, 768, 1599, 2304, 1, 1, 1),
    ("hubert fc1", 1, 768, 1599, 3072, 1, 1, 1),
    ("hubert fc2", 1, 3072, 1599, 768, 1, 1, 1),
    ("enc_p ffn1", 1, 192, 3198, 768, 3, 1, 1),
    ("flow in k5", 1, 192, 3198, 384, 5, 1, 1),
    ("gemm 4096", 1, 4096, 4096, 4096, 1, 1, 1),
]
if __name__ == "__main__":
    ctx = _lib.Context(0)
    for name, B, Cin, Tin, Cout, K, s, d in SHAPES:
        ms, tf = ctx.bench_conv1d(B, Cin, Tin, Cout, K, s, d, 1, 5)
        print(f"{name:22s} {ms:9.3f} ms  {tf:7.1f} TFLOP/s  ({tf / 157.3 * 100:4.1f} % of fp32 MFMA peak)")
