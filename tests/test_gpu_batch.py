"""Batch / multi-model / determinism properties of rvcx_convert_batch (BASELINE configs C3 and C5 in small):
ragged batches reproduce each utterance run alone (bit-exact), two voice models stay resident side by side,
repeated runs are bit-identical (deterministic split-K and stream joins), forced conv tiles agree."""
import numpy as np
import pytest

from conftest import rms

pytestmark = pytest.mark.gpu


def _load(ctx, seed, synth_cfgs):
    from polgen_rvc_amd import synthetic as S, weights as W
    hcfg, rcfg = S.HUBERT_CFG_TINY, S.RMVPE_CFG_TINY
    ctx.load_hubert(W.hubert_cfg_struct(hcfg), S.hubert_state(hcfg, seed))
    ctx.load_rmvpe(W.rmvpe_cfg_struct(rcfg), S.rmvpe_state(rcfg, seed))
    mids = []
    for i, scfg in enumerate(synth_cfgs):
        st = S.synth_state(scfg, seed + 10 * i, input_dim=hcfg["embed_dim"])
        mids.append(ctx.load_synth(W.synth_cfg_struct(scfg, hcfg["embed_dim"]), st))
    return mids


def _params(index_rate=0.0, protect=0.33, seed=5, volume_envelope=1.0):
    from polgen_rvc_amd import _lib
    return _lib.Params(0.0, 50.0, 1100.0, index_rate, protect, volume_envelope, 0, 1, 1, 2, 3, seed)


def test_ragged_batch_with_index_equals_single(ctx):
    """C3 in small: B = 3 clips of different lengths (one long enough to be cut into chunks), retrieval blend
    at index_rate 0.75 with protect 0.33, Philox noise: the batch call returns exactly what each clip gives alone."""
    from polgen_rvc_amd import synthetic as S
    (mid,) = _load(ctx, 3, [S.SYNTH_CFG_TINY])
    big = S.make_index(2048, S.HUBERT_CFG_TINY["embed_dim"], 1)
    ctx.load_index(big)
    try:
        clips = [S.make_clip(40, 1.7), S.make_clip(41, 5.3), S.make_clip(42, 2.9)]
        p = _params(index_rate=0.75, volume_envelope=0.25)
        batch = ctx.convert_batch(mid, clips, p)
        # utterance i of a call draws from Philox(seed + i): its single run uses that seed
        singles = [ctx.convert_batch(mid, [c], _params(index_rate=0.75, volume_envelope=0.25, seed=5 + i))[0]
                   for i, c in enumerate(clips)]
        assert [len(b) for b in batch] == [len(a) for a in singles]
        for alone, b in zip(singles, batch):
            assert np.array_equal(alone, b)
        # the blend is live: without the index the output differs
        p0 = _params(index_rate=0.0, volume_envelope=0.25)
        assert not np.array_equal(ctx.convert_batch(mid, [clips[0]], p0)[0], batch[0])
    finally:
        ctx.load_index(None)


def test_two_resident_voice_models(ctx):
    """C5 in small: two synthesizers (different weights and upsampling) share one HuBERT + RMVPE in a context;
    interleaved calls give the same PCM as each model used on its own, and sample counts follow each model's rate."""
    from polgen_rvc_amd import synthetic as S
    cfg_b = list(S.SYNTH_CFG_TINY)
    cfg_b[12], cfg_b[14], cfg_b[17] = [5, 2, 2, 2], [9, 4, 4, 4], 4000      # upp 40 -> 4 kHz
    m_a, m_b = _load(ctx, 6, [S.SYNTH_CFG_TINY, cfg_b])
    clip = S.make_clip(50, 2.2)
    p = _params()
    a1 = ctx.convert_batch(m_a, [clip], p)[0]
    b1 = ctx.convert_batch(m_b, [clip], p)[0]
    a2 = ctx.convert_batch(m_a, [clip], p)[0]
    b2 = ctx.convert_batch(m_b, [clip], p)[0]
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2)
    assert len(a1) * 4000 == len(b1) * 4800
    assert ctx.synth_upp(m_a) == 48 and ctx.synth_upp(m_b) == 40


def test_repeat_is_bit_identical_and_seed_matters(ctx):
    from polgen_rvc_amd import synthetic as S
    (mid,) = _load(ctx, 8, [S.SYNTH_CFG_TINY])
    clip = S.make_clip(60, 4.0)
    r1, f1 = ctx.convert_batch(mid, [clip], _params(seed=11), want_f32=True)
    r2, f2 = ctx.convert_batch(mid, [clip], _params(seed=11), want_f32=True)
    r3, _ = ctx.convert_batch(mid, [clip], _params(seed=12), want_f32=True)
    assert np.array_equal(r1[0], r2[0]) and np.array_equal(f1[0], f2[0])
    assert not np.array_equal(r1[0], r3[0])
    assert np.isfinite(f1[0]).all() and rms(f1[0]) > 1e-4


@pytest.mark.parametrize("shape", [(1, 128, 3000, 128, 7, 1, 3), (1, 768, 700, 192, 1, 1, 1), (1, 512, 2001, 512, 3, 2, 1)])
def test_forced_tiles_and_splitk_agree(ctx, shape):
    """Every tile of the compile-time family and every split-K factor computes the same conv: forced
    configurations (rvcx_conv_override) agree with the heuristic's choice to fp32 rounding (1e-6 relative)."""
    import torch
    import torch.nn.functional as F
    B, Cin, T, Cout, K, s, d = shape
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, T, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    pad = (K * d - d) // 2 if s == 1 else 0
    ref = F.conv1d(x, w, b, stride=s, dilation=d, padding=pad).numpy()
    try:
        base = ctx.conv1d(x.numpy(), w.numpy(), b.numpy(), stride=s, dil=d, pad_left=pad, Tout=ref.shape[2])
        assert rms(base - ref) / rms(ref) < 2e-6
        seen = 0
        for tile in range(15):
            for sk in (1, 2, 4):
                ctx.conv_override(tile, 0, sk)
                got = ctx.conv1d(x.numpy(), w.numpy(), b.numpy(), stride=s, dil=d, pad_left=pad, Tout=ref.shape[2])
                assert rms(got - ref) / rms(ref) < 2e-6, (tile, sk)
                seen += 1
        assert seen == 45
        # fp16 hi/lo split kernels (conv_h3.hip): fp32-grade products, same tolerance
        for tile in (100, 101, 102):
            ctx.conv_override(tile, 0, 1)
            got = ctx.conv1d(x.numpy(), w.numpy(), b.numpy(), stride=s, dil=d, pad_left=pad, Tout=ref.shape[2])
            assert rms(got - ref) / rms(ref) < 2e-6, tile
    finally:
        ctx.conv_override(-1, -1, -1)
