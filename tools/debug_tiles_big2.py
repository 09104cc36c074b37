#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib
ctx = _lib.Context(0)
g = torch.Generator().manual_seed(0)
B, C, H, W = 3, 16, 3232, 128
x = torch.randn(B, C, H, W, generator=g).numpy()
w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).numpy()
ref = np.concatenate([ctx.conv2d3x3(x[b:b + 1], w, None, act=2) for b in range(B)])
for tile in (-1, 110, -1, 110, 109, -1):
    ctx.conv_override(tile, -1, -1)
    got = ctx.conv2d3x3(x, w, None, act=2)
    bad = np.argwhere(got != ref)
    print(f"tile {tile}: equal {np.array_equal(got, ref)} nbad {len(bad)} first bad {bad[:3].tolist()} last bad {bad[-2:].tolist()}", flush=True)
