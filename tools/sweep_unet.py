import os, sys
sys.path.insert(0, '/root/repo')
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib
ctx = _lib.Context(0)
for name, cin, T, cout in [("L0 C16", 16, 420160, 16), ("L1 C32", 32, 106656, 32), ("L2 C64", 64, 27472, 64)]:
    for t in (-1, 103, 104, 109, 110):
        ctx.conv_override(t, 0, 1)
        ms, tf = ctx.bench_conv1d(1, cin, T, cout, 9, 1, 32, 1, 20)
        print(name, "tile", t, f"{ms*1e3:.1f} us")
