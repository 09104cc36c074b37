"""conv_ws (csrc/conv_deep.hip: the weight-stationary 64 x 320 tile, round 6) against the conv_h3 tiles it replaces, per
shape, B = 1 and B = 16: HIP-event time per launch through the library's own profile hooks (a split launch includes its
finish kernel).  usage (GPU box): python tools/bench_ws.py [1d]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RVCX_DEBUG", "1")
import numpy as np
import polgen_rvc_amd  # noqa: F401
from polgen_rvc_amd import _lib

ctx = _lib.Context(0)
g = np.random.default_rng(0)


def timed(fn, n=10):
    fn()
    ctx.conv_profile_begin()
    for _ in range(n):
        fn()
    pr = ctx.conv_profile_end()
    ms = sum(p["ms"] for p in pr) / n
    return ms * 1e3, "+".join(sorted({p["tile"].split("(")[0].strip() for p in pr}))


def row(tag, fn, gflop):
    out = []
    for name, tile in (("ws", 163), ("h3", -1 if os.environ.get("RVCX_CONV_WS") == "0" else 104)):
        try:
            ctx.conv_override(tile, -1, -1)
            us, tiles = timed(fn)
        finally:
            ctx.conv_override(-1, -1, -1)
        out.append(f"{name} {us:7.1f} us {gflop / us * 1e3:6.1f} TF/s [{tiles}]")
    print(f"{tag:38s} " + "   ".join(out), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "1d":
    # 1-D layers of the synthesizer: (Cin, Cout, k, dil, T)
    for (ci, co, k, d, T) in [(256, 256, 7, 1, 38376), (256, 256, 11, 1, 38376), (256, 256, 11, 5, 38376), (256, 256, 3, 1, 38376),
                              (192, 512, 7, 1, 3198), (192, 768, 3, 1, 3326), (768, 192, 3, 1, 3326), (192, 384, 5, 1, 3326),
                              (128, 128, 7, 1, 383760), (64, 64, 11, 1, 767520)]:
        for B in (1, 8):
            x = g.standard_normal((B, ci, T)).astype(np.float32)
            w = (g.standard_normal((co, ci, k)) / np.sqrt(k * ci)).astype(np.float32)
            pad = (k - 1) * d // 2
            row(f"1d B={B} {ci}->{co} k{k} d{d} T={T}", lambda: ctx.conv1d(x, w, None, dil=d, pad_left=pad, pre_lrelu=0.1),
                2.0 * B * co * T * k * ci * 1e-9)
            if T > 100000:
                break
else:
    for (ci, co, H, W) in [(512, 512, 101, 4), (256, 512, 101, 4), (256, 256, 202, 8), (512, 256, 202, 8), (128, 128, 404, 16),
                           (256, 128, 404, 16), (64, 64, 808, 32), (128, 64, 808, 32)]:
        for B in (1, 16):
            x = g.standard_normal((B, ci, H, W)).astype(np.float32)
            w = (g.standard_normal((co, ci, 3, 3)) / np.sqrt(9 * ci)).astype(np.float32)
            row(f"3x3 B={B} {ci}->{co} {H}x{W}", lambda: ctx.conv2d3x3(x, w, None, act=2), 2.0 * B * co * H * W * 9 * ci * 1e-9)
    for (ci, co, H, W) in [(512, 256, 101, 4), (256, 128, 202, 8), (128, 64, 404, 16)]:
        for B in (1, 16):
            x = g.standard_normal((B, ci, H, W)).astype(np.float32)
            w = (g.standard_normal((ci, co, 3, 3)) / np.sqrt(9 * ci / 4)).astype(np.float32)
            row(f"convT2d B={B} {ci}->{co} {H}x{W}", lambda: ctx.convtranspose2d(x, w, None, act=2), 2.0 * B * co * H * W * 9 * ci * 1e-9)
