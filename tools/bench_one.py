#!/usr/bin/env python3
"""Run one conv shape a few times (for rocprofv3 --pmc runs): bench_one.py B Cin Tin Cout K stride dil iters"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib
a = [int(v) for v in sys.argv[1:9]]
ctx = _lib.Context(0)
ms, tf = ctx.bench_conv1d(a[0], a[1], a[2], a[3], a[4], a[5], a[6], 1, a[7])
print(f"{ms:.3f} ms {tf:.1f} TFLOP/s")
