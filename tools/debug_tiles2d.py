#!/usr/bin/env python3
"""Every 3x3 tile of the conv_h3 family against torch conv2d on U-Net-like shapes, B = 1 and 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib
ctx = _lib.Context(0)
def rms(a): return float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))
g = torch.Generator().manual_seed(0)
for (B, C, H, W) in [(1, 16, 96, 128), (4, 16, 96, 128), (1, 64, 48, 32), (4, 64, 48, 32), (4, 256, 24, 8), (3, 128, 101, 16)]:
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b = torch.randn(C, generator=g)
    ref = F.relu(F.conv2d(x, w, b, padding=1)).numpy()
    for tile in (-1, 103, 104, 109, 110):
        ctx.conv_override(tile, -1, -1)
        try:
            got = ctx.conv2d3x3(x.numpy(), w.numpy(), b.numpy(), act=2)
            e = rms(got - ref) / rms(ref)
            print(f"B={B} C={C} H={H} W={W} tile {tile}: rel err {e:.2e} finite {np.isfinite(got).all()}", flush=True)
        except Exception as ex:
            print(f"B={B} C={C} H={H} W={W} tile {tile}: {str(ex)[:100]}", flush=True)
ctx.conv_override(-1, -1, -1)
