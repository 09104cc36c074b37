#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib
ctx = _lib.Context(0)
g = torch.Generator().manual_seed(0)
for (B, C, H, W) in [(3, 16, 3232, 128), (3, 32, 1616, 64), (4, 64, 808, 32), (3, 128, 404, 16)]:
    x = torch.randn(B, C, H, W, generator=g).numpy()
    w = (torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5).numpy()
    res = torch.randn(B, C, H, W, generator=g).numpy()
    ctx.conv_override(-1, -1, -1)
    ref = np.concatenate([ctx.conv2d3x3(x[b:b + 1], w, None, res=res[b:b + 1], act=2) for b in range(B)])
    for tile in (-1, 103, 104, 109, 110):
        ctx.conv_override(tile, -1, -1)
        got = ctx.conv2d3x3(x, w, None, res=res, act=2)
        print(f"B={B} C={C} H={H} W={W} tile {tile}: equal single {np.array_equal(got, ref)} finite {np.isfinite(got).all()} "
              f"maxdiff {np.nanmax(np.abs(got - ref)):.2e}", flush=True)
ctx.conv_override(-1, -1, -1)
