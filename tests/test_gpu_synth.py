"""Synthesizer.infer on the GPU (C-ABI rvcx_synth_infer) vs the committed reference goldens and
the CPU oracle.  Floating point: fp32 everywhere; tolerance stated per test."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import rms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_layernorm_c(ctx):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 192, 301, generator=g) * 3 + 0.5
    gamma, beta = torch.randn(192, generator=g), torch.randn(192, generator=g)
    ref = torch.nn.functional.layer_norm(x.transpose(1, 2), (192,), gamma, beta, 1e-5).transpose(1, 2)
    got = ctx.layernorm_c(x.numpy(), gamma.numpy(), beta.numpy())
    assert rms(got - ref.numpy()) / rms(ref.numpy()) < 1e-5


@pytest.mark.parametrize("case", [(1, 2, 96, 300, True), (2, 12, 64, 257, False), (1, 2, 24, 37, True),
                                  (1, 4, 32, 1000, False)])
def test_attention(ctx, case):
    """softmax(q k^T * scale [+ rel-pos k]) v [+ rel-pos v], attentions.py:63-113 (tolerance 2e-5 rel)."""
    B, H, D, T, rel = case
    g = torch.Generator().manual_seed(7)
    q, k, v = (torch.randn(B, H * D, T, generator=g) for _ in range(3))
    ek = torch.randn(21, D, generator=g) * D ** -0.5 if rel else None
    ev = torch.randn(21, D, generator=g) * D ** -0.5 if rel else None
    scale = D ** -0.5
    qh = q.view(B, H, D, T).transpose(2, 3) * scale
    kh = k.view(B, H, D, T).transpose(2, 3)
    vh = v.view(B, H, D, T).transpose(2, 3)
    sc = qh @ kh.transpose(-1, -2)
    if rel:
        rl = qh @ ek.t()
        for r in range(21):
            off = r - 10
            if abs(off) < T:
                lo, hi = max(0, -off), T - max(0, off)
                sc.diagonal(offset=off, dim1=-2, dim2=-1).add_(rl[..., lo:hi, r])
    p = torch.softmax(sc, -1)
    out = p @ vh
    if rel:
        band = torch.zeros(B, H, T, 21)
        for r in range(21):
            off = r - 10
            if abs(off) < T:
                lo, hi = max(0, -off), T - max(0, off)
                band[..., lo:hi, r] = p.diagonal(offset=off, dim1=-2, dim2=-1)
        out = out + band @ ev
    ref = out.transpose(2, 3).reshape(B, H * D, T).numpy()
    got = ctx.attention(q.numpy(), k.numpy(), v.numpy(), H, scale, None if ek is None else ek.numpy(),
                        None if ev is None else ev.numpy())
    e = rms(got - ref) / rms(ref)
    assert e < 2e-5, e


def _load_model(ctx, cfg, seed):
    from polgen_rvc_amd import synthetic as S, weights as W
    sd = S.synth_state(cfg, seed)
    return ctx.load_synth(W.synth_cfg_struct(cfg, 768), sd), sd


@pytest.mark.parametrize("tag", ["tiny", "48k_T24"])
def test_synth_vs_reference_golden(ctx, tag):
    """Inputs/outputs captured from the reference's Synthesizer.infer (tools/gen_golden.py).
    Tolerance: 1e-4 relative RMS on the fp32 waveform (north-star budget is 1e-3 absolute)."""
    d = np.load(os.path.join(GOLD, f"synth_{tag}.npz"))
    cfg = json.loads(str(d["cfg"]))
    mid, _ = _load_model(ctx, cfg, int(d["seed"]))
    got, stats, zflow = ctx.synth_infer(mid, d["phone"], d["pitch"], d["f0"], z_noise=d["z_noise"],
                                        src_noise=d["src_noise"][:, :, 0], taps=True)
    # the intermediates the reference returns beside the waveform: TextEncoder statistics and the flow output
    inter = d["m_p"].shape[1]
    for name, mine, ref_t in (("m_p", stats[:, :inter], d["m_p"]), ("logs_p", stats[:, inter:], d["logs_p"]),
                              ("z", zflow, d["z"])):
        et = rms(mine - ref_t) / max(rms(ref_t), 1e-12)
        print(f"synth {tag}: {name} rel err {et:.3e}")
        assert mine.shape == ref_t.shape and et < 1e-4, name
    ref = d["audio"][:, 0]
    e = rms(got - ref)
    print(f"synth {tag}: rms_ref={rms(ref):.4f} rms_err={e:.3e}")
    assert np.isfinite(got).all()
    assert e / rms(ref) < 1e-4 and e < 1e-4


def test_synth_ragged_batch_equals_single(ctx):
    """B=3 with different lengths must reproduce each item run alone (the reference is B=1 only)."""
    from polgen_rvc_amd import synthetic as S
    from oracle import synth as O
    cfg = S.SYNTH_CFG_TINY
    mid, sd = _load_model(ctx, cfg, 3)
    c = O.cfg_fields(cfg)
    g = torch.Generator().manual_seed(1)
    T, lens = 50, [50, 33, 17]
    phone = torch.randn(3, T, 768, generator=g)
    pitch = torch.randint(1, 256, (3, T), generator=g)
    f0 = 100 + 300 * torch.rand(3, T, generator=g)
    f0[:, 5:9] = 0
    zn = torch.randn(3, c["inter"], T, generator=g)
    sn = torch.randn(3, T * c["upp"], generator=g)
    got = ctx.synth_infer(mid, phone.numpy(), pitch.numpy(), f0.numpy(), lens=lens, z_noise=zn.numpy(),
                          src_noise=sn.numpy())
    sdt = S.to_torch(sd)
    for i, L in enumerate(lens):
        ref = O.synthesizer_infer(sdt, cfg, phone[i:i + 1, :L], torch.tensor([L]), pitch[i:i + 1, :L],
                                  f0[i:i + 1, :L], torch.tensor([0]), zn[i:i + 1, :, :L],
                                  sn[i:i + 1, :L * c["upp"], None])[0, 0].numpy()
        e = rms(got[i, :L * c["upp"]] - ref) / rms(ref)
        assert e < 1e-4, (i, e)


def test_synth_philox_noise_runs(ctx):
    """Production mode (no noise handed in): on-device Philox draws; output finite, seed-reproducible."""
    from polgen_rvc_amd import synthetic as S
    mid, _ = _load_model(ctx, S.SYNTH_CFG_TINY, 3)
    g = torch.Generator().manual_seed(2)
    phone = torch.randn(1, 40, 768, generator=g).numpy()
    pitch = np.full((1, 40), 60)
    f0 = np.full((1, 40), 220.0, np.float32)
    a = ctx.synth_infer(mid, phone, pitch, f0, seed=5)
    b = ctx.synth_infer(mid, phone, pitch, f0, seed=5)
    c2 = ctx.synth_infer(mid, phone, pitch, f0, seed=6)
    assert np.isfinite(a).all() and (a == b).all() and not (a == c2).all()


def test_synth_48k_ragged_batch_equals_single(ctx):
    """Full-size 48 k synthesizer (its ResBlock convs run on the split-fp16 kernels, c1 -> c2 hand-off in split
    form): a ragged batch (B = 2, 40 and 23 frames) must reproduce each item run alone to fp32 rounding."""
    from polgen_rvc_amd import synthetic as S
    cfg = S.SYNTH_CFG_48K
    mid, _ = _load_model(ctx, cfg, 7)
    g = torch.Generator().manual_seed(2)
    T, lens = 40, [40, 23]
    upp = 480
    phone = torch.randn(2, T, 768, generator=g)
    pitch = torch.randint(1, 256, (2, T), generator=g)
    f0 = 100 + 300 * torch.rand(2, T, generator=g)
    zn = torch.randn(2, 192, T, generator=g)
    sn = torch.randn(2, T * upp, generator=g)
    both = ctx.synth_infer(mid, phone.numpy(), pitch.numpy(), f0.numpy(), lens=lens, z_noise=zn.numpy(),
                           src_noise=sn.numpy())
    for i, L in enumerate(lens):
        one = ctx.synth_infer(mid, phone[i:i + 1, :L].numpy(), pitch[i:i + 1, :L].numpy(), f0[i:i + 1, :L].numpy(),
                              z_noise=zn[i:i + 1, :, :L].numpy().copy(), src_noise=sn[i:i + 1, :L * upp].numpy().copy())
        e = rms(both[i, :L * upp] - one[0, :L * upp]) / rms(one[0, :L * upp])
        assert np.isfinite(both[i]).all() and e < 1e-5, (i, e)
        if L < T:
            assert np.abs(both[i, L * upp:]).max() == 0.0      # beyond the item's length: exact zeros
