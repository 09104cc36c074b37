import os, sys
sys.path.insert(0, "/root/repo")
import polgen_rvc_amd
from polgen_rvc_amd import _lib
ctx = _lib.Context(0)
for name, c, T, d in [("L0 C16", 16, 3232*130, 33), ("L1 C32", 32, 1616*66, 17), ("L0dec 32->16", 32, 3232*130, 33)]:
    cout = 16 if "dec" in name else c
    ctx.conv_override(-1, -1, -1)
    base, tf = ctx.bench_conv1d(1, c, T, cout, 9, 1, d, 1, 20)
    out = [f"heuristic {base*1e3:.1f} us ({tf:.1f} TF/s)"]
    for t in (103, 109, 110):
        ctx.conv_override(t, 0, 1)
        try:
            ms, tf = ctx.bench_conv1d(1, c, T, cout, 9, 1, d, 1, 20)
            out.append(f"[tile {t-100}: {ms*1e3:.1f} us]")
        except Exception as e:
            out.append(f"[tile {t-100}: {str(e)[:30]}]")
    io = (c + 2 * cout) * T * 4 / 1e6
    print(name, " ".join(out), f"| compulsory I/O {io:.0f} MB = {io/4.5e3*1e3:.1f} us at 4.5 TB/s", flush=True)
