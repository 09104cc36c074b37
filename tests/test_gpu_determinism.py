"""Per-kernel determinism under a co-resident load.

Round 3 met a BiGRU step whose result depended on what else shared the compute units: with the time-major GEMM of a
second context running beside it, the cluster kernel returned different F0 tracks for identical inputs (DESIGN.md
"A 16-byte LDS store that lost a dword"; the in-tree switch -DRVCX_GRU_B128=1 + tools/check_gru_under_load.py keep the
failing form alive).  The idiom it used -- 16-byte LDS stores, a raw `s_waitcnt lgkmcnt(0); s_barrier`, several
workgroups per CU -- is the backbone of gemm_h3, resblock_pair, conv_h3 and attn_h3, and the batch == single contract
rests on every one of them giving the same bits every time.  Here each of them (and the BiGRU cluster kernel) runs
REPS times on one input while a second context keeps the chip busy from another host thread with (a) the time-major
GEMM, (b) the fused ResBlock step: exactly one distinct result is allowed."""
import hashlib
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPS = 60


def _victims(ctx):
    from polgen_rvc_amd import synthetic as S
    g = np.random.Generator(np.random.PCG64(1))
    x1 = g.standard_normal((1, 384, 3232)).astype(np.float32)
    w1 = (g.standard_normal((1536, 384, 1)) / 20).astype(np.float32)
    x2 = g.standard_normal((1, 192, 3198)).astype(np.float32)
    w2 = (g.standard_normal((768, 192, 3)) / 24).astype(np.float32)
    x3 = g.standard_normal((1, 64, 20000)).astype(np.float32)
    w3 = (g.standard_normal((64, 64, 7)) / 21).astype(np.float32)
    b3 = g.standard_normal(64).astype(np.float32)
    x3b = g.standard_normal((1, 128, 12000)).astype(np.float32)
    w3b = (g.standard_normal((128, 128, 11)) / 37).astype(np.float32)
    b3b = g.standard_normal(128).astype(np.float32)
    x4 = g.standard_normal((1, 768, 1599)).astype(np.float32)
    w4 = (g.standard_normal((2304, 768)) / 28).astype(np.float32)
    q = g.standard_normal((1, 768, 1599)).astype(np.float32)
    sd = {k: v for k, v in S.rmvpe_state(S.RMVPE_CFG_FULL, 1900).items() if k.startswith("fc.0.gru")}
    xg = (0.5 * g.standard_normal((1, 3232, 384))).astype(np.float32)
    return [
        ("conv_h3 k=1 384->1536 T=3232", lambda: ctx.conv1d(x1, w1)),
        ("conv_h3 k=3 192->768 T=3198", lambda: ctx.conv1d(x2, w2, pad_left=1)),
        ("resblock_pair C=64 k=7 d=3", lambda: ctx.resblock_pair(x3, w3, b3, w3, b3, dil=3)),
        ("resblock_pair C=128 k=11 d=5", lambda: ctx.resblock_pair(x3b, w3b, b3b, w3b, b3b, dil=5)),
        ("gemm_h3 (time-major) 768->2304 T=1599", lambda: ctx.gemm_tm(x4, w4)[0]),
        ("attn_h3 12 x 64 T=1599", lambda: ctx.attention(q, q * 0.5, q * 0.25, 12, 0.125)),
        ("bigru (cluster of 4 workgroups per direction) T=3232", lambda: ctx.bigru(xg, sd)),
    ]


@pytest.mark.parametrize("load", ["gemm", "pair"])
def test_every_hot_kernel_gives_one_result_under_a_second_contexts_load(ctx, load):
    from polgen_rvc_amd import _lib
    other = _lib.Context(0)
    stop = threading.Event()
    err = []

    def background():
        try:
            while not stop.is_set():
                if load == "gemm":
                    other.bench_gemm(1599, 768, 3072, 50)
                else:
                    other.bench_resblock_pair(1, 128, 383760, 7, 3, True, 3)
        except Exception as e:  # noqa: BLE001
            err.append(e)

    th = threading.Thread(target=background)
    th.start()
    try:
        fallbacks0 = ctx.gru_fallbacks()
        report = []
        for name, fn in _victims(ctx):
            digests = {hashlib.sha256(np.ascontiguousarray(fn()).tobytes()).hexdigest() for _ in range(REPS)}
            report.append((name, len(digests)))
        print(f"load = {load}: " + "; ".join(f"{n}: {d} distinct of {REPS}" for n, d in report))
        assert all(d == 1 for _, d in report), report
        assert ctx.gru_fallbacks() == fallbacks0        # the cluster kernel itself ran (no time-out fallback)
    finally:
        stop.set()
        th.join()
        other.close()
    assert not err, err
