#!/usr/bin/env python3
"""Are different tiles of one family bit-identical?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib
ctx = _lib.Context(0)
g = torch.Generator().manual_seed(0)
for (B, C, H, W) in [(1, 64, 48, 32), (3, 128, 101, 16), (1, 16, 404, 128)]:
    x = torch.randn(B, C, H, W, generator=g).numpy()
    w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).numpy()
    outs = {}
    for tile in (103, 104, 109, 110):
        ctx.conv_override(tile, -1, 1)
        outs[tile] = ctx.conv2d3x3(x, w, None, act=2)
    print(f"2D B={B} C={C} H={H} W={W}: bit-equal to tile 103:", {t: bool(np.array_equal(outs[103], o)) for t, o in outs.items()}, flush=True)
for (B, C, T, K, d) in [(1, 128, 3000, 7, 3), (2, 64, 5000, 11, 5), (1, 32, 9000, 3, 1)]:
    x = torch.randn(B, C, T, generator=g).numpy()
    w = (torch.randn(C, C, K, generator=g) / (C * K) ** 0.5).numpy()
    outs = {}
    for tile in (100, 101, 102):
        ctx.conv_override(tile, -1, 1)
        outs[tile] = ctx.conv1d(x, w, None, dil=d, pad_left=(K * d - d) // 2)
    print(f"1D B={B} C={C} T={T} k={K}: bit-equal to tile 100:", {t: bool(np.array_equal(outs[100], o)) for t, o in outs.items()}, flush=True)
ctx.conv_override(-1, -1, -1)
