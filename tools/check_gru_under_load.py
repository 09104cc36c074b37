#!/usr/bin/env python3
"""Is the BiGRU cluster kernel deterministic while other kernels share the GPU? (debugging aid)
A second context keeps the chip busy with GEMM launches from another thread; the main thread runs rvcx_op_bigru on
the same input N times and counts distinct results.
The round-3 failure lives in the round-3 kernel: run the -DRVCX_GRU_B128=1 build with RVCX_GRU_FORM=0."""
import hashlib, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("RVCX_DEBUG", "1")   # tuning hooks are refused without it
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib, synthetic as S
load = sys.argv[1] if len(sys.argv) > 1 else "gemm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx, ctx2 = _lib.Context(0), _lib.Context(0)
sd = {k: v for k, v in S.rmvpe_state(S.RMVPE_CFG_FULL, 1900).items() if k.startswith("fc.0.gru")}
g = np.random.Generator(np.random.PCG64(1))
x = (0.5 * g.standard_normal((1, 3232, 384))).astype(np.float32)
stop = False


def bg():
    while not stop:
        if load == "gemm":
            ctx2.bench_gemm(1599, 768, 3072, 50)
        elif load == "pair":
            ctx2.bench_resblock_pair(1, 128, 383760, 7, 3, True, 3)


th = None
if load != "none":
    th = threading.Thread(target=bg)
    th.start()
outs = []
words = []
for _ in range(n):
    outs.append(ctx.bigru(x, sd))
    words.append(_lib.lib().rvcx_debug_inject(ctx._h, 2))
print("device error words per run:", words)
stop = True
if th:
    th.join()
hs = [hashlib.sha256(o.tobytes()).hexdigest()[:8] for o in outs]
d = [float(np.abs(o - outs[0]).max()) for o in outs]
print(f"load={load}: distinct {len(set(hs))} of {n}; max abs diff vs first {max(d):.2e}; fallbacks {ctx.gru_fallbacks()}")
for i, o in enumerate(outs[1:], 1):
    w = np.nonzero((o != outs[0]).any(axis=2)[0])[0]
    if len(w):
        print(f"  run {i}: {len(w)} frames differ, first {w[:8].tolist()}, dirs fwd {bool((o[0, :, :256] != outs[0][0, :, :256]).any())} rev {bool((o[0, :, 256:] != outs[0][0, :, 256:]).any())}")
# forward direction: the first step at which a run differs from run 0 -- which units, by how much?
for i, o in enumerate(outs[1:], 1):
    fw = (o[0, :, :256] != outs[0][0, :, :256])
    st = np.nonzero(fw.any(axis=1))[0]
    if len(st):
        s0 = int(st[0])
        u = np.nonzero(fw[s0])[0]
        d = np.abs(o[0, s0, :256] - outs[0][0, s0, :256])
        print(f"  run {i} fwd: first differing step {s0}: {len(u)} units {u[:10].tolist()} (wg slices {sorted(set((u // 64).tolist()))}), "
              f"max diff {d.max():.3e}, values {o[0, s0, u[:3]].tolist()} vs {outs[0][0, s0, u[:3]].tolist()}")
    rv = (o[0, :, 256:] != outs[0][0, :, 256:])
    st = np.nonzero(rv.any(axis=1))[0]
    if len(st):
        s0 = int(st[-1])
        u = np.nonzero(rv[s0])[0]
        d = np.abs(o[0, s0, 256:] - outs[0][0, s0, 256:])
        print(f"  run {i} rev: first differing step {3231 - s0} (frame {s0}): {len(u)} units {u[:10].tolist()} (wg slices {sorted(set((u // 64).tolist()))}), max diff {d.max():.3e}")
