"""Round-6 parity additions (VERDICT r5 "What's weak" 1 and 2):

* the cut-point / per-chunk / trim / concat branch of VC.pipeline (pipeline.py:330-344,381-447) at the REAL geometry
  (1,6,38,41) and model size, against the reference's own output on a 95 s clip (3 chunks) -- until now that branch was
  pinned to the reference only at the toy geometry (1,1,2,3);
* a 240 s stress clip (no golden: output length = the reference's formula, finite, batch == single, fast path);
* NSF decoder and F0 U-Net with planted outlier channels (synthetic.synth_state / rmvpe_state(outliers=True)) against
  the reference's Synthesizer.infer / E2E on the same weights, and BASELINE C2 end to end with them.
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import FULL_PCM_BAR, FULL_RMS_BAR, rms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _setup(ctx, cfgs, seed, dec_outliers=False, unet_outliers=False):
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I
    hcfg, rcfg, scfg = cfgs
    I._CTX[0] = ctx
    hub = I.load_hubert("cuda:0", False, None, state=S.hubert_state(hcfg, seed), cfg=hcfg)
    I.load_rmvpe("cuda:0", state=S.rmvpe_state(rcfg, seed, outliers=unet_outliers), cfg=rcfg)
    cpt = S.synth_checkpoint(scfg, seed)
    cpt["weight"] = S.synth_state(scfg, seed, input_dim=hcfg["embed_dim"], outliers=dec_outliers)
    return hub, cpt


def _chunk_noise(d, cfgs, tgt_sr):
    """The reference's Gaussian draws, regenerated from the recorded seed of the private generator the harness fed it with
    (tools/gen_golden.py: run_ref_pipeline), in its draw order: per chunk z (1, inter, T) then the source noise (1, T upp, 1)."""
    gen = torch.Generator().manual_seed(int(d["noise_seed"]))
    upp = tgt_sr // 100
    parts = []
    for L in d["chunk_lens"]:
        T = int(L) // upp
        z = torch.randn((1, cfgs[2][2], T), generator=gen)
        src = torch.randn((1, T * upp, 1), generator=gen)
        parts += [z.numpy().ravel(), src.numpy().ravel()]
    return np.concatenate(parts)


def _trim_map(chunk_lens, tp):
    """index of every sample of the concatenated UN-trimmed chunk outputs in the trimmed, concatenated waveform (-1: cut off)"""
    out, base = [], 0
    for L in chunk_lens:
        L = int(L)
        m = np.full(L, -1, np.int64)
        m[tp:L - tp] = base + np.arange(L - 2 * tp)
        out.append(m)
        base += L - 2 * tp
    return np.concatenate(out), base


def _compare_strided(tag, pcm, f32, d, tgt_sr, pcm_bar=FULL_PCM_BAR, frac_bar=0.02):
    tp = tgt_sr * int(d["geo"][0])
    tmap, n_out = _trim_map(d["chunk_lens"], tp)
    assert len(pcm) == n_out == len(f32), (len(pcm), n_out)
    ref_pcm = d["pcm_samples"].astype(np.int32)
    assert len(pcm[::997]) == len(ref_pcm)
    diff = np.abs(pcm[::997].astype(np.int32) - ref_pcm)
    idx = np.arange(0, len(tmap), 997)
    keep = tmap[idx] >= 0
    e = rms(f32[tmap[idx[keep]]] - d["raw_samples"][keep])
    msg = (f"{tag}: float rms err {e:.3e} (bar {FULL_RMS_BAR:g}; north star 1e-3; signal rms {rms(d['raw_samples']):.3f}), "
           f"pcm max diff {diff.max()} LSB (bar {pcm_bar}), frac>1 {np.mean(diff > 1):.2e} (bar {frac_bar})")
    print(msg)
    assert e < FULL_RMS_BAR, msg
    assert diff.max() <= pcm_bar and np.mean(diff > 1) < frac_bar, msg
    # every sample: RMS of each 4096-sample block of the reference's un-trimmed chunk outputs that lies inside one
    # chunk's kept region
    ref_b, checked = d["block_rms"], 0
    for b in range(len(ref_b)):
        lo, hi = b * 4096, min(len(tmap), (b + 1) * 4096)
        m = tmap[lo:hi]
        if (m < 0).any() or (np.diff(m) != 1).any():
            continue
        got = rms(f32[m[0]: m[-1] + 1])
        assert abs(got - float(ref_b[b])) <= 2e-4 * max(1.0, float(ref_b[b])) + 5e-6, (b, got, float(ref_b[b]))
        checked += 1
    return checked


def test_95s_three_chunks_full_size_vs_reference_golden(ctx):
    """A 95 s clip at (1,6,38,41), 48 k, rmvpe+, through the reference's VC.pipeline (tools/gen_golden.py --full --only
    pipe_long95): F0 once over ~9 700 frames (a 9 700-step BiGRU), three silence-aligned chunks, per-chunk noise, trim and
    concatenation.  The noise layout handed in follows the REFERENCE's chunk lengths, so a cut point that differed by one
    frame would misalign every draw behind it; the output length is checked against the reference's."""
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I
    d = np.load(os.path.join(GOLD, "pipeline_long95_48k.npz"))
    cfgs = json.loads(str(d["cfgs"]))
    assert int(d["n_chunks"]) >= 3
    hub, cpt = _setup(ctx, cfgs, int(d["seed"]))
    cpt, version, net_g, tgt_sr, vc = I.get_vc("cuda:0", False, I.Config(), None, cpt=cpt)
    audio = S.make_clip(int(d["clip"]), float(d["seconds"]))
    noise = _chunk_noise(d, cfgs, tgt_sr)
    fb0 = ctx.gru_fallbacks()
    pcm, f32 = vc.pipeline(hub, net_g, 0, audio, "x.wav", 0.0, "rmvpe+", None, 0, 1, 3, tgt_sr, 0, 1.0, "v2", 0.33,
                           128, None, 50, 1100, noise=noise, return_f32=True)
    # Bars: the float waveform keeps C2's RMS bar (3e-5; measured 2.0e-5).  The PCM bar is NOT C2's 4 LSB: the NSF sine
    # source integrates f0 over a whole chunk without a reset (nsf.py / generators.py: cumsum of f0 / sr), so the GPU's
    # f0 rounding noise (<= 1e-6 relative, asserted below) random-walks into ~1e-3 rad at the end of a 39 s chunk --
    # single samples there differ by up to 11 LSB (3.4e-4 of full scale; 3 % of the samples by more than 1 LSB), a 32 s
    # chunk (C2) stays within 2.  The reference itself moves like this between BLAS builds; the bar is 16 LSB / 8 %.
    nblk = _compare_strided("long95", pcm, f32, d, tgt_sr, pcm_bar=16, frac_bar=0.08)
    assert nblk >= len(pcm) // 4096 - 3 * int(d["n_chunks"])
    assert ctx.gru_fallbacks() == fb0
    x = np.pad(ctx.highpass(audio.astype(np.float64)), (vc.t_pad, vc.t_pad), mode="reflect")
    coarse, f0 = vc.get_f0("x", x, len(d["f0"]), 0.0, "rmvpe+", 3, 128, None, 50, 1100)
    assert np.mean(coarse[:len(d["f0"])] != d["coarse"]) < 1e-3
    v = (d["f0"] > 0) & (f0[:len(d["f0"])] > 0)
    rel = np.abs(f0[:len(d["f0"])][v] - d["f0"][v]) / d["f0"][v]
    print(f"long95: f0 max rel err {rel.max():.2e} over {int(v.sum())} voiced frames")
    assert rel.max() < 1e-5 and np.array_equal(d["f0"] > 0, f0[:len(d["f0"])] > 0)
    print(f"long95: chunks {d['chunk_lens'].tolist()}, stage ms {ctx.last_timing()}")


def test_240s_stress_clip(ctx):
    """A 4 min song-length clip (the reference's real workload: 6 chunks, F0 once over 24 200 frames): finite output of the
    length the reference's arithmetic gives (every chunk boundary is a multiple of the 10 ms window: n // 160 * upp), the
    same PCM inside a batch of two as alone, no BiGRU fallback, no layer pinned to fp32."""
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I
    cfgs = (S.HUBERT_CFG_BASE, S.RMVPE_CFG_FULL, S.SYNTH_CFG_48K)
    hub, cpt = _setup(ctx, cfgs, 0)
    cpt, version, net_g, tgt_sr, vc = I.get_vc("cuda:0", False, I.Config(), None, cpt=cpt)
    audio = S.make_clip(77, 240.0)
    fb0, l0 = ctx.gru_fallbacks(), ctx.fp32_layers()
    vc.seed = 5
    pcm = vc.pipeline(hub, net_g, 0, audio, "x.wav", 0.0, "rmvpe+", None, 0, 1, 3, tgt_sr, 0, 1.0, "v2", 0.33,
                      128, None, 50, 1100)
    t = ctx.last_timing()
    # pipeline.py:381-447: every chunk is cut at multiples of the 10 ms window and loses at most the two frames HuBERT's
    # framing drops (a 30 s clip: 3198 of 3200 frames); a clip of this length has at most 7 chunks
    upp = tgt_sr // 100
    short = (len(audio) // 160) * upp - len(pcm)
    assert pcm.dtype == np.int16 and 0 <= short <= 2 * upp * 7 and short % upp == 0, short
    assert np.abs(pcm.astype(np.int32)).max() > 1000 and rms(pcm) > 100
    # no silent stretch of 1 s anywhere (a dropped chunk would be one)
    blk = pcm[: len(pcm) // tgt_sr * tgt_sr].reshape(-1, tgt_sr).astype(np.float64)
    assert (np.sqrt((blk ** 2).mean(1)) > 10).all()
    vc.seed = 5
    two = vc.pipeline_batch(hub, net_g, 0, [audio, audio], 0.0, "rmvpe+", None, 0, 1, tgt_sr, 0, 1.0, "v2", 0.33,
                            None, 50, 1100)
    # utterance i of a batch draws its Gaussians from Philox(seed + i) (include/rvcx.h): item 1 equals a single run at seed + 1
    vc.seed = 6
    pcm1 = vc.pipeline(hub, net_g, 0, audio, "x.wav", 0.0, "rmvpe+", None, 0, 1, 3, tgt_sr, 0, 1.0, "v2", 0.33,
                       128, None, 50, 1100)
    assert (two[0] == pcm).all() and (two[1] == pcm1).all() and not (pcm1 == pcm).all()
    assert ctx.gru_fallbacks() == fb0 and ctx.fp32_layers() == l0
    print(f"240 s clip: {len(pcm)} samples, stage ms {t}")


def test_synth_with_decoder_outliers_vs_reference_golden(ctx):
    """Synthesizer.infer of the reference on a 48 k model whose NSF decoder carries planted outlier channels (weight-norm g
    spread 1 : 150 in ups.0, c1 rows x 150 / c2 columns / 150 in one step of every ResBlock1): same bars as the clean
    golden.  Prints how many layers the fp16-range guard pinned to the exact-fp32 kernels (expected 0: values of a few
    hundred are inside the split kernels' range)."""
    from polgen_rvc_amd import synthetic as S, weights as W
    d = np.load(os.path.join(GOLD, "synth_48k_T24_outliers.npz"))
    cfg = json.loads(str(d["cfg"]))
    assert bool(d["outliers"])
    l0, r0 = ctx.fp32_layers(), ctx.fp32_reruns()
    mid = ctx.load_synth(W.synth_cfg_struct(cfg, 768), S.synth_state(cfg, int(d["seed"]), outliers=True))
    got, stats, zflow = ctx.synth_infer(mid, d["phone"], d["pitch"], d["f0"], z_noise=d["z_noise"],
                                        src_noise=d["src_noise"][:, :, 0], taps=True)
    ez = rms(zflow - d["z"]) / rms(d["z"])
    ref = d["audio"][:, 0]
    e = rms(got - ref)
    print(f"synth 48k with decoder outliers: z rel err {ez:.3e}, audio rms_ref={rms(ref):.4f} rms_err={e:.3e}; "
          f"layers pinned to fp32 by the range guard: {ctx.fp32_layers() - l0}, repeated calls: {ctx.fp32_reruns() - r0}")
    assert ez < 1e-4 and np.isfinite(got).all()
    assert e / rms(ref) < 1e-4 and e < 1e-4
    assert ctx.fp32_layers() == l0, "activations of a few hundred must stay on the split-fp16 kernels"


def test_rmvpe_with_unet_outliers_vs_reference_golden(ctx):
    """E2E of the reference with a BatchNorm scale of 150 planted on channels of eight ConvBlockRes hand-offs (one per
    encoder level, intermediate, two decoder blocks): salience within 1e-4 relative of the reference, f0 within 1e-3."""
    from polgen_rvc_amd import synthetic as S, weights as W
    d = np.load(os.path.join(GOLD, "rmvpe_full_1s_outliers.npz"))
    cfg = json.loads(str(d["cfg"]))
    assert bool(d["outliers"])
    l0 = ctx.fp32_layers()
    ctx.load_rmvpe(W.rmvpe_cfg_struct(cfg), S.rmvpe_state(cfg, int(d["seed"]), outliers=True))
    f0, hid = ctx.rmvpe_f0(d["audio"], return_hidden=True)
    st = int(d["stride"])
    e = rms(hid[0, ::st] - d["hidden"]) / rms(d["hidden"])
    print(f"rmvpe with U-Net outliers: hidden rel err {e:.3e}; layers pinned to fp32: {ctx.fp32_layers() - l0}")
    assert e < 1e-4
    ref = d["f0"]
    both = (ref > 0) & (f0[0] > 0)
    assert (np.abs(f0[0][both] - ref[both]) / ref[both]).max() < 1e-3
    assert ((ref > 0) != (f0[0] > 0)).mean() < 0.01
    assert ctx.fp32_layers() == l0


def test_c2_with_outlier_decoder_and_unet_vs_reference_golden(ctx):
    """BASELINE C2 end to end with the outlier-planted decoder AND U-Net.  The planting is function-preserving in exact
    arithmetic (synthetic.py), so the reference's golden of the clean model is the golden of this one up to fp32 rounding
    (the reference itself moves by 8e-8 RMS on the synthesizer, 3e-7 on the salience): same bars as the clean C2 test."""
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I
    d = np.load(os.path.join(GOLD, "pipeline_c2_30s_48k.npz"))
    cfgs = json.loads(str(d["cfgs"]))
    hub, cpt = _setup(ctx, cfgs, int(d["seed"]), dec_outliers=True, unet_outliers=True)
    l0, r0 = ctx.fp32_layers(), ctx.fp32_reruns()
    cpt, version, net_g, tgt_sr, vc = I.get_vc("cuda:0", False, I.Config(), None, cpt=cpt)
    audio = S.make_clip(int(d["clip"]), float(d["seconds"]))
    noise = _chunk_noise(d, cfgs, tgt_sr)
    pcm, f32 = vc.pipeline(hub, net_g, 0, audio, "x.wav", 0.0, "rmvpe+", None, 0, 1, 3, tgt_sr, 0, 1.0, "v2", 0.33,
                           128, None, 50, 1100, noise=noise, return_f32=True)
    _compare_strided("c2 with decoder + U-Net outliers", pcm, f32, d, tgt_sr)
    print(f"layers pinned to fp32 by the range guard: {ctx.fp32_layers() - l0}, repeated calls: {ctx.fp32_reruns() - r0}\n"
          + ctx.fp32_pinned())
    assert ctx.fp32_layers() - l0 <= 1, "at most one layer may leave the split kernels on this model"


# ---------------------------------------------------------------- conv_ws (csrc/conv_deep.hip): the weight-stationary tile
@pytest.mark.parametrize("B,Cin,Cout,H,W,S", [(1, 512, 512, 101, 4, 8), (1, 256, 512, 101, 4, 4), (2, 256, 256, 202, 8, 8),
                                              (1, 128, 128, 404, 16, 4), (1, 64, 64, 808, 32, 2), (3, 128, 64, 100, 32, 2)])
def test_weight_stationary_tile_equals_the_64x64_tile_bit_for_bit(ctx, B, Cin, Cout, H, W, S):
    """conv_ws (64 x 320 tile, a stage = one 16-channel chunk x all taps) keeps conv_h3's k-order inside a K segment and its
    segment bounds: with the same number of segments the two kernels agree bit for bit (conv_h3 forced through the override
    with split-K = S; conv_ws told to cut K into S segments), and both agree with torch within fp32 rounding."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B * 100 + Cin + Cout)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=g)
    res = torch.randn(B, Cout, H, W, generator=g)
    ref = (F.relu(F.conv2d(x, w, bias, padding=1)) + res).numpy()
    args = (x.numpy(), w.numpy(), bias.numpy())
    try:
        ctx.conv_override(104 if 2 * (W + 2) + 2 > 64 else 102, -1, S)      # the 64 x 64 tile whose halo class the map's width falls in
        ctx.conv_profile_begin()
        tiled = ctx.conv2d3x3(*args, res=res.numpy(), act=2)
        assert all(r["tile"].startswith("conv_h3<64,64") for r in ctx.conv_profile_end())
        ctx.conv_override(163, -1, S)
        ctx.conv_profile_begin()
        got = ctx.conv2d3x3(*args, res=res.numpy(), act=2)
        names = [r["tile"] for r in ctx.conv_profile_end()]
    finally:
        ctx.conv_override(-1, -1, -1)
    assert any(n.startswith("conv_ws") for n in names), names
    e = rms(got - ref) / rms(ref)
    print(f"B={B} {Cin}->{Cout} {H}x{W} S={S}: vs torch {e:.2e}; equal to the 64 x 64 tile: {np.array_equal(got, tiled)}")
    assert np.isfinite(got).all() and e < 2e-6
    assert np.array_equal(got, tiled)


@pytest.mark.parametrize("Cin,Cout,H,W", [(512, 512, 101, 4), (256, 256, 202, 8), (256, 512, 101, 4), (512, 256, 202, 8)])
def test_weight_stationary_tile_batch_equals_single(ctx, Cin, Cout, H, W):
    """The default path of the F0 U-Net's 3 x 3 convs with >= 256 input channels.  A single item cuts K into S segments over S
    workgroups + the finish kernel; a batch of 8 walks the same segments inside one workgroup and adds them in the same
    order: every item of the batch equals its single run bit for bit (ragged row counts included), and torch within fp32
    rounding."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(Cin + H)
    B = 8
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5
    bias = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv2d(x, w, bias, padding=1)).numpy()
    ctx.conv_profile_begin()
    batch = ctx.conv2d3x3(x.numpy(), w.numpy(), bias.numpy(), act=2)
    names = [r["tile"] for r in ctx.conv_profile_end()]
    assert any(n.startswith("conv_ws") for n in names), names
    assert rms(batch - ref) / rms(ref) < 2e-6
    for i in (0, 5, 7):
        alone = ctx.conv2d3x3(x[i:i + 1].numpy(), w.numpy(), bias.numpy(), act=2)
        assert np.array_equal(alone[0], batch[i]), i


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(1, 512, 256, 101, 4), (4, 256, 128, 50, 8)])
def test_polyphase_convtranspose2d_on_the_weight_stationary_tile(ctx, B, Cin, Cout, H, W):
    """ConvTranspose2d(3, stride 2) of the U-Net decoder (RMVPE.py:262-281) as a four-phase conv with 2 x 2 taps + shuffle
    store: through conv_ws (split + finish for one item, segments in one workgroup for a batch) against torch."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(Cin + W)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, 3, 3, generator=g) / (Cin * 9 / 4) ** 0.5
    bias = torch.randn(Cout, generator=g)
    ref = F.relu(F.conv_transpose2d(x, w, bias, stride=2, padding=1, output_padding=1)).numpy()
    try:
        ctx.conv_override(163, -1, -1)         # not a default shape of conv_ws (measured slower): forced, the form stays tested
        ctx.conv_profile_begin()
        got = ctx.convtranspose2d(x.numpy(), w.numpy(), bias.numpy(), act=2)
        names = [r["tile"] for r in ctx.conv_profile_end()]
        assert any(n.startswith("conv_ws") for n in names), names
        assert np.isfinite(got).all() and rms(got - ref) / rms(ref) < 2e-6
        if B > 1:
            alone = ctx.convtranspose2d(x[1:2].numpy(), w.numpy(), bias.numpy(), act=2)
            assert np.array_equal(alone[0], got[1])
    finally:
        ctx.conv_override(-1, -1, -1)


@pytest.mark.parametrize("B,C,H,W,ragged", [(1, 256, 202, 8, False), (3, 256, 100, 8, True), (2, 512, 40, 4, True)])
def test_convblockres_of_the_deep_levels_through_the_models_block_path(ctx, B, C, H, W, ragged):
    """One ConvBlockRes (RMVPE.py:140-175) of the levels conv_ws serves, through the F0 model's own block path, per-item
    row counts as in a ragged micro-batch: against torch per item at its own height."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B * 7 + C)
    x = torch.randn(B, C, H, W, generator=g)
    w1 = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    w2 = torch.randn(C, C, 3, 3, generator=g) / (C * 9) ** 0.5
    b1, b2 = torch.randn(C, generator=g), torch.randn(C, generator=g)
    rows = np.array([H, H // 2 + 3, 5][:B], np.int32) if ragged else None
    ref = torch.zeros(B, C, H, W)
    for b in range(B):
        hb = int(rows[b]) if ragged else H
        xb = x[b:b + 1, :, :hb]
        t = F.relu(F.conv2d(xb, w1, b1, padding=1))
        ref[b, :, :hb] = F.relu(F.conv2d(t, w2, b2, padding=1)) + xb
    ctx.conv_profile_begin()
    got = ctx.convblock2d(x.numpy(), w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy(), rows=rows)
    names = [r["tile"] for r in ctx.conv_profile_end()]
    assert sum(n.startswith("conv_ws") for n in names) >= 1, names
    e = rms(got - ref.numpy()) / rms(ref.numpy())
    print(f"B={B} C={C} {H}x{W} ragged={ragged}: vs torch {e:.2e}")
    assert np.isfinite(got).all() and e < 2e-6
    if B > 1:
        rb = None if rows is None else rows[1:2]
        alone = ctx.convblock2d(x[1:2].numpy(), w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy(), rows=rb)
        assert np.array_equal(alone[0], got[1])


# ---------------------------------------------------------------- RVC v1 voice models (VERDICT r5 "What's missing" 4)
@pytest.mark.parametrize("tag", ["tiny_v1", "v1_5s_40k"])
def test_v1_voice_model_vs_reference_golden(ctx, tag):
    """An RVC v1 voice model through the reference's VC.pipeline(version="v1") (tools/gen_golden.py): HuBERT output layer 9
    (not 12) + ``final_proj`` (768 -> 256 at full size), the retrieval / protect / synthesizer path on the narrower features
    (rvc/infer/pipeline.py:228-236, rvc/infer/infer.py:91-97).  Tiny configs: every sample; full size (HuBERT-base, 40 k
    synthesizer, 5 s): the strided fixture with C1's bars."""
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I
    d = np.load(os.path.join(GOLD, f"pipeline_{tag}.npz"))
    assert str(d["version"]) == "v1"
    hcfg, rcfg, scfg = json.loads(str(d["cfgs"]))
    seed = int(d["seed"])
    I._CTX[0] = ctx
    hub = I.load_hubert("cuda:0", False, None, state=S.hubert_state(hcfg, seed), cfg=hcfg)
    I.load_rmvpe("cuda:0", state=S.rmvpe_state(rcfg, seed), cfg=rcfg)
    cpt = S.synth_checkpoint(scfg, seed, version="v1")
    cpt["weight"] = S.synth_state(scfg, seed, input_dim=hcfg["final_dim"])
    cpt, version, net_g, tgt_sr, vc = I.get_vc("cuda:0", False, I.Config(), None, cpt=cpt)
    assert version == "v1" and net_g.input_dim == hcfg["final_dim"]
    audio = S.make_clip(int(d["clip"]), float(d["seconds"]))
    args = (hub, net_g, 0, audio, "x.wav", float(d["pitch"]), "rmvpe+", None, 0, 1, 3, tgt_sr, 0, float(d["volume_envelope"]))
    tail = (float(d["protect"]), 128, None, float(d["f0_min"]), float(d["f0_max"]))
    if "pcm" in d.files:                       # tiny: every sample and the un-trimmed float output
        noise = np.concatenate([np.concatenate([d[f"z_noise_{i}"].ravel(), d[f"src_noise_{i}"].ravel()])
                                for i in range(int(d["n_chunks"]))]).astype(np.float32)
        pcm, f32 = vc.pipeline(*args, "v1", *tail, noise=noise, return_f32=True)
        tp = tgt_sr * int(d["geo"][0])
        raw = d["raw"][tp:len(d["raw"]) - tp]
        e = rms(f32 - raw)
        diff = np.abs(pcm.astype(np.int32) - d["pcm"].astype(np.int32))
        print(f"{tag}: float rms err {e:.3e} (rms {rms(raw):.3f}); pcm max diff {diff.max()} LSB")
        assert pcm.shape == d["pcm"].shape and e < 2e-5 and diff.max() <= FULL_PCM_BAR
    else:
        noise = _chunk_noise(d, (hcfg, rcfg, scfg), tgt_sr)
        pcm, f32 = vc.pipeline(*args, "v1", *tail, noise=noise, return_f32=True)
        _compare_strided(tag, pcm, f32, d, tgt_sr)
    # the caller's version must agree with the model (the reference would die inside emb_phone with a shape error)
    if hcfg["final_dim"] == 256:
        with pytest.raises(ValueError, match="does not match"):
            vc.pipeline(*args, "v2", *tail)


def test_bigru_publish_probe_holds_on_this_device(ctx):
    """The cluster BiGRU's hand-off inside one XCD is a plain store + sc1 poll (gru.hip): a hardware observation, checked once
    per device by gru_publish_probe_kernel when RMVPE is loaded.  On the MI355X of this pool it must hold (1: the fast publish
    is the one the F0 goldens run); RVCX_EXPECT_PROBE=0 is the forced-failure run of tests/test_gpu_modes.py, where the state
    must read 0 and the recurrence must still complete without a time-out fallback."""
    from polgen_rvc_amd import synthetic as S, weights as W
    want = int(os.environ.get("RVCX_EXPECT_PROBE", "1"))
    cfg = S.RMVPE_CFG_TINY
    ctx.load_rmvpe(W.rmvpe_cfg_struct(cfg), S.rmvpe_state(cfg, 3))
    assert ctx.gru_publish_probe() == want
    fb0 = ctx.gru_fallbacks()
    rng = np.random.default_rng(5)
    f0 = ctx.rmvpe_f0(rng.standard_normal(16000).astype(np.float32) * 0.1)
    assert np.all(np.isfinite(f0)) and ctx.gru_fallbacks() == fb0


def _window_inputs(cfg, E, B, T, seed):
    g = np.random.default_rng(seed)
    inter, upp = cfg[2], int(np.prod(cfg[12]))
    phone = (g.standard_normal((B, T, E)) * 0.3).astype(np.float32)
    pitchf = (160.0 + 60.0 * np.sin(np.arange(T)[None] / 9.0 + g.uniform(0, 3, (B, 1)))).astype(np.float32)
    pitchf[:, T // 3:T // 3 + 20] = 0.0                          # an unvoiced stretch
    pitch = np.clip(np.rint(1 + 254 * (pitchf - 50) / 1050), 1, 255).astype(np.int32)
    zn = g.standard_normal((B, inter, T)).astype(np.float32)
    sn = g.standard_normal((B, T * upp)).astype(np.float32)
    return phone, pitch, pitchf, zn, sn, upp


@pytest.mark.parametrize("name,T", [("48k", 264), ("40k", 250), ("tiny", 300)])
def test_decoder_window_gives_the_samples_of_the_full_evaluation(ctx, name, T):
    """SynthIO::dec_skip (round 6): the NSF decoder evaluated on frames [skip, T - skip) only must return, inside what
    VC.pipeline keeps (audio1[t_pad_tgt:-t_pad_tgt], t_pad = 100 frames), the samples of the full evaluation -- up to fp32
    rounding where a launch's tile / split choice follows the shorter length (measured: 0 or ~1e-7) -- and silence outside
    its window.  Negative control: a window that starts only 4 frames in front of the kept samples (inside the receptive
    field, tests/test_decoder_window.py) must NOT reproduce their first frames."""
    from polgen_rvc_amd import synthetic as S, weights as W
    cfg = {"48k": S.SYNTH_CFG_48K, "40k": S.SYNTH_CFG_40K, "tiny": S.SYNTH_CFG_TINY}[name]
    E = 768 if name != "tiny" else S.HUBERT_CFG_TINY["embed_dim"]
    mid = ctx.load_synth(W.synth_cfg_struct(cfg, E), S.synth_state(cfg, 11, input_dim=E))
    try:
        rf = ctx.synth_dec_rf(mid)
        assert rf == W.synth_dec_rf(cfg)
        pad = 100
        skip = (pad - rf) & ~3
        phone, pitch, pitchf, zn, sn, upp = _window_inputs(cfg, E, 1, T, 3)
        full = ctx.synth_infer(mid, phone, pitch, pitchf, z_noise=zn, src_noise=sn)[0]
        win = ctx.synth_infer(mid, phone, pitch, pitchf, z_noise=zn, src_noise=sn, dec_skip=skip)[0]
        keep = slice(pad * upp, (T - pad) * upp)
        d = np.abs(win[keep] - full[keep]).max()
        print(f"{name}: rf {rf}, skip {skip} of {T} frames; kept region max |diff| {d:.2e} (signal rms {rms(full[keep]):.3f})")
        assert d <= 2e-6
        assert not win[:skip * upp].any() and not win[(T - skip) * upp:].any()
        assert np.abs(full[:skip * upp]).max() > 1e-3                    # (what was skipped was not silence)
        bad = ctx.synth_infer(mid, phone, pitch, pitchf, z_noise=zn, src_noise=sn, dec_skip=pad - 4)[0]
        first = slice(pad * upp, (pad + 2) * upp)
        assert np.abs(bad[first] - full[first]).max() > 1e-4, "the control window is inside the receptive field"
    finally:
        ctx.unload_synth(mid)


def test_decoder_window_in_a_ragged_group(ctx):
    """members of different lengths: each is windowed at its own length ([skip, len - skip)), results inside each member's
    kept region equal its full evaluation"""
    from polgen_rvc_amd import synthetic as S, weights as W
    cfg, E = S.SYNTH_CFG_TINY, S.HUBERT_CFG_TINY["embed_dim"]
    mid = ctx.load_synth(W.synth_cfg_struct(cfg, E), S.synth_state(cfg, 12, input_dim=E))
    try:
        T, lens, pad = 320, [320, 276, 301], 100
        skip = (pad - ctx.synth_dec_rf(mid)) & ~3
        phone, pitch, pitchf, zn, sn, upp = _window_inputs(cfg, E, 3, T, 4)
        full = ctx.synth_infer(mid, phone, pitch, pitchf, lens=lens, z_noise=zn, src_noise=sn)
        win = ctx.synth_infer(mid, phone, pitch, pitchf, lens=lens, z_noise=zn, src_noise=sn, dec_skip=skip)
        for b, L in enumerate(lens):
            keep = slice(pad * upp, (L - pad) * upp)
            assert np.abs(win[b, keep] - full[b, keep]).max() <= 2e-6, b
            assert not win[b, :skip * upp].any() and not win[b, (L - skip) * upp:].any()
    finally:
        ctx.unload_synth(mid)
