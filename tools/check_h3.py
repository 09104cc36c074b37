#!/usr/bin/env python3
"""conv_h3 (fp16 hi/lo split MFMA) vs torch fp32/fp64 and vs the fp32-MFMA path: error and speed."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RVCX_DEBUG", "1")   # tuning hooks are refused without it
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib
ctx = _lib.Context(0)
rms = lambda a: float(np.sqrt(np.mean(np.asarray(a, np.float64) ** 2)))
for (Cin, T, Cout, K, d) in [(32, 5000, 32, 3, 1), (64, 4001, 64, 7, 3), (128, 3000, 128, 11, 5), (256, 999, 256, 7, 1), (48, 1500, 80, 5, 2)]:
    g = torch.Generator().manual_seed(Cin + K)
    x = torch.randn(1, Cin, T, generator=g) * 3
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    pad = (K * d - d) // 2
    ref64 = F.conv1d(F.leaky_relu(x.double(), 0.1), w.double(), b.double(), dilation=d, padding=pad).numpy()
    ref32 = F.conv1d(F.leaky_relu(x, 0.1), w, b, dilation=d, padding=pad).numpy()
    ctx.conv_override(-1, -1, -1)
    os.environ["X"] = "1"
    f32 = ctx.conv1d(x.numpy(), w.numpy(), b.numpy(), dil=d, pad_left=pad, Tout=T, pre_lrelu=0.1)
    out = [f"fp32-mfma {rms(f32 - ref64) / rms(ref64):.2e}", f"torch-f32 {rms(ref32 - ref64) / rms(ref64):.2e}"]
    for t in (100, 101, 102):
        ctx.conv_override(t, 0, 1)
        got = ctx.conv1d(x.numpy(), w.numpy(), b.numpy(), dil=d, pad_left=pad, Tout=T, pre_lrelu=0.1)
        out.append(f"h3[{t-100}] {rms(got - ref64) / rms(ref64):.2e} max {np.abs(got - ref64).max():.2e}")
    print((Cin, T, Cout, K, d), " | ".join(out))
ctx.conv_override(-1, -1, -1)
