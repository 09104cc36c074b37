#!/usr/bin/env python3
"""Determinism of single ops while a second context keeps the GPU busy from another thread (debugging aid)."""
import hashlib, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib, synthetic as S
load = sys.argv[1] if len(sys.argv) > 1 else "gemm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx, ctx2 = _lib.Context(0), _lib.Context(0)
g = np.random.Generator(np.random.PCG64(1))
stop = False


def bg():
    while not stop:
        if load == "gemm":
            ctx2.bench_gemm(1599, 768, 3072, 50)
        elif load == "conv":
            ctx2.bench_conv1d(1, 768, 1599, 3072, 1, iters=50)
        elif load == "pair":
            ctx2.bench_resblock_pair(1, 128, 383760, 7, 3, True, 3)


def distinct(name, fn):
    outs = [fn() for _ in range(n)]
    hs = {hashlib.sha256(np.ascontiguousarray(o).tobytes()).hexdigest()[:8] for o in outs}
    d = max(float(np.abs(o - outs[0]).max()) for o in outs)
    print(f"  {name:34s} distinct {len(hs)} of {n}   max abs diff {d:.2e}", flush=True)


th = None
if load != "none":
    th = threading.Thread(target=bg)
    th.start()
print(f"load = {load}")
x1 = g.standard_normal((1, 384, 3232)).astype(np.float32)
w1 = (g.standard_normal((1536, 384, 1)) / 20).astype(np.float32)
distinct("conv1d k=1 384->1536 T=3232", lambda: ctx.conv1d(x1, w1))
x2 = g.standard_normal((1, 192, 3198)).astype(np.float32)
w2 = (g.standard_normal((768, 192, 3)) / 24).astype(np.float32)
distinct("conv1d k=3 192->768 T=3198", lambda: ctx.conv1d(x2, w2, pad_left=1))
x3 = g.standard_normal((1, 64, 20000)).astype(np.float32)
w3 = (g.standard_normal((64, 64, 7)) / 21).astype(np.float32)
b3 = g.standard_normal(64).astype(np.float32)
distinct("resblock_pair C=64 k=7", lambda: ctx.resblock_pair(x3, w3, b3, w3, b3, dil=3))
x4 = g.standard_normal((1, 768, 1599)).astype(np.float32)
w4 = (g.standard_normal((2304, 768)) / 28).astype(np.float32)
distinct("gemm_tm 768->2304 T=1599", lambda: ctx.gemm_tm(x4, w4)[0])
q = g.standard_normal((1, 768, 1599)).astype(np.float32)
distinct("attention 12x64 T=1599", lambda: ctx.attention(q, q * 0.5, q * 0.25, 12, 0.125))
x5 = g.standard_normal((1, 768, 1599)).astype(np.float32)
distinct("layernorm_c 768 x 1599", lambda: ctx.layernorm_c(x5, np.ones(768), np.zeros(768)))
sd = {k: v for k, v in S.rmvpe_state(S.RMVPE_CFG_FULL, 1900).items() if k.startswith("fc.0.gru")}
xg = (0.5 * g.standard_normal((1, 3232, 384))).astype(np.float32)
distinct("bigru (cluster) T=3232", lambda: ctx.bigru(xg, sd))
os.environ["X"] = "1"
stop = True
if th:
    th.join()
