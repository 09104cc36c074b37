#!/usr/bin/env python3
"""Micro-benchmark of the time-major Linear kernel (csrc/gemm.hip) on the HuBERT-base shapes of a 30 s clip (rows =
1599 per item) for every tile, B = 1 and B = 8.  usage: bench_gemm.py [iters=30]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RVCX_DEBUG", "1")   # tuning hooks are refused without it
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib

SHAPES = [("qkv", 768, 2304), ("o", 768, 768), ("fc1", 768, 3072), ("fc2", 3072, 768)]
TILES = ["128x128", "64x128", "128x64", "64x64", "bd128x256"]
if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    ctx = _lib.Context(0)
    for B in (1, 8, 16):
        rows = 1599 * B
        tot_auto = 0.0
        for name, cin, cout in SHAPES:
            line = f"B={B} {name:4s} {cin:4d}->{cout:4d}: "
            for t, tn in enumerate(TILES):
                _lib.Context.conv_override(tile=200 + t)
                ms, tf = ctx.bench_gemm(rows, cin, cout, iters)
                line += f"{tn} {ms * 1e3:6.1f} us {tf:5.0f} TF/s | "
            _lib.Context.conv_override()
            ms, tf = ctx.bench_gemm(rows, cin, cout, iters)
            tot_auto += ms
            print(line + f"auto {ms * 1e3:6.1f} us {tf:5.0f} TF/s ({tf / 833.3 * 100:4.1f} %)", flush=True)
        # the long-K layer (fc2): K summed in four segments by one workgroup per tile (splitk=1) or by four (splitk=2)
        for sk, what in ((1, "one workgroup per tile"), (2, "K-split x4 + finish")):
            line = f"B={B} fc2 {what:24s}: "
            for t, tn in enumerate(TILES):
                _lib.Context.conv_override(tile=200 + t, splitk=sk)
                ms, tf = ctx.bench_gemm(rows, 3072, 768, iters)
                line += f"{tn} {ms * 1e3:6.1f} us | "
            _lib.Context.conv_override()
            print(line, flush=True)
        print(f"B={B}: one layer's four GEMMs {tot_auto * 1e3:.1f} us; x 12 layers = {tot_auto * 12:.2f} ms")
