"""FCPE F0 back-end (rvcx_load_fcpe / rvcx_fcpe_f0 / rvcx_get_f0_fcpe_x, VC.get_f0(f0_method="fcpe")) on the GPU
against the vectors captured from the reference's own FCPE module and VC.get_f0 / VC.pipeline call sites
(tests/golden/fcpe_*.npz, pipeline_*fcpe*.npz; tools/gen_golden.py) and against the CPU oracle.  fp32;
tolerances stated per test."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import rms

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(ctx, cfg, seed):
    from polgen_rvc_amd import synthetic as S, weights as W
    sd = S.fcpe_state(cfg, seed)
    ctx.load_fcpe(W.fcpe_cfg_struct(W.fcpe_cfg_from_state(sd)), sd)
    return sd


@pytest.mark.parametrize("tag", ["tiny", "full_2s"])
def test_fcpe_stages_vs_reference_golden(ctx, tag):
    """log-mel (slaney basis, sqrt(re^2+im^2+1e-9), repeated last frame) <= 1e-4 rel; sigmoid salience of
    FCPE.forward <= 1e-4 rel RMS; decoded Hz <= 1e-4 rel on voiced frames with identical voicing decisions
    (the fixture's seed keeps every salience maximum >= 0.2 % away from the 0.03 threshold)."""
    d = np.load(os.path.join(GOLD, f"fcpe_{tag}.npz"))
    cfg = json.loads(str(d["cfg"]))
    _load(ctx, cfg, int(d["seed"]))
    st = int(d["stride"])
    f0, sal, mel = ctx.fcpe_f0(d["x"], 0.03, return_salience=True, return_mel=True)
    assert f0.shape == (1, len(d["x"]) // 160 + 1)
    em = rms(mel[0][:, ::st] - d["mel"]) / rms(d["mel"])
    es = rms(sal[0][::st] - d["salience"]) / rms(d["salience"])
    ref = d["raw_f0"]
    print(f"fcpe {tag}: log-mel rel err {em:.3e}; salience rel err {es:.3e} (max abs "
          f"{np.abs(sal[0][::st] - d['salience']).max():.2e}); voiced {int((f0 > 0).sum())}/{f0.size}")
    assert mel[0][:, ::st].shape == d["mel"].shape and em < 1e-4
    assert es < 1e-4
    assert ((ref > 0) == (f0[0] > 0)).all()
    v = ref > 0
    assert (np.abs(f0[0][v] - ref[v]) / ref[v]).max() < 1e-4


@pytest.mark.parametrize("tag", ["tiny", "full_2s"])
def test_get_f0_fcpe_vs_reference_golden(ctx, tag):
    """VC.get_f0(..., "fcpe", ...) as the reference returns it (pipeline.py:169-201): p_len frames, unvoiced frames
    bridged by np.interp in float64, pitch shift, coarse.  f0 <= 1e-4 rel (max), coarse identical."""
    from polgen_rvc_amd.infer import infer as I
    d = np.load(os.path.join(GOLD, f"fcpe_{tag}.npz"))
    cfg = json.loads(str(d["cfg"]))
    _load(ctx, cfg, int(d["seed"]))
    ctx.fcpe_loaded = True
    I._CTX[0] = ctx
    vc = I.VC(48000, I.Config())
    p_len = len(d["x"]) // 160
    coarse, f0 = vc.get_f0("x.wav", d["x"], p_len, float(d["pitch"]), "fcpe", 3, 128, None, 50, 1100)
    assert coarse.shape == f0.shape == (p_len,) and coarse.dtype == np.int64 and f0.dtype == np.float64
    e = (np.abs(f0 - d["f0"]) / np.maximum(d["f0"], 1.0)).max()
    print(f"get_f0 fcpe {tag}: f0 max rel err {e:.3e}; coarse differs at {int((coarse != d['coarse']).sum())} frames")
    assert e < 1e-4
    assert (coarse == d["coarse"]).all()


def test_fcpe_post_process_vs_oracle_edge_cases(ctx):
    """FCPEF0Predictor.post_process + get_f0's tail through the device kernel on hand-made raw tracks: nothing
    voiced, a single voiced frame, voiced runs touching either end, long unvoiced gaps, p_len equal to / smaller
    than / larger than the model's frame count (nearest resize in float32 index arithmetic), 10 minutes of frames
    (np.interp in float64 on the reference's two differently rounded time axes, 512*i/16000 vs 0.032*i).
    f0 must be float32-exact against the oracle, coarse identical."""
    from oracle import fcpe as OF, pipeline as OP
    rng = np.random.default_rng(3)

    def track(n, voiced_frac, runs=True):
        f = (100.0 + 400.0 * rng.random(n)).astype(np.float32)
        if runs:
            gate = np.repeat(rng.random(n // 7 + 1) < voiced_frac, 7)[:n]
        else:
            gate = rng.random(n) < voiced_frac
        return np.where(gate, f, 0).astype(np.float32)

    cases = [np.zeros(301, np.float32), track(301, 0.5), track(3001, 0.3), track(3001, 0.9, False), track(60001, 0.4)]
    one = np.zeros(301, np.float32); one[123] = 222.5
    ends = track(501, 0.5); ends[0] = 0; ends[1] = 150.0; ends[-1] = 0; ends[-2] = 0; ends[-3] = 310.0
    first_last = track(501, 0.5); first_last[0] = 99.0; first_last[-2] = 500.0
    cases += [one, ends, first_last]
    for raw in cases:
        n = len(raw)
        for p_len in sorted({n - 1, n, max(1, n // 3), n + 17}):
            coarse, f0 = ctx.fcpe_post(raw, p_len, pitch=-5.0)
            ref = OF.post_process(raw, p_len) if (raw != 0).any() else np.zeros(p_len)
            rc, rf = OP.f0_to_coarse(ref, -5.0, 50, 1100)
            assert np.array_equal(f0, rf.astype(np.float32)), (n, p_len, np.abs(f0 - rf).max())
            assert np.array_equal(coarse, rc), (n, p_len)


def test_fcpe_batch_equals_single(ctx):
    """B = 3 signals through one launch sequence: every row bit-identical to converting it alone."""
    from polgen_rvc_amd import synthetic as S
    _load(ctx, S.FCPE_CFG_TINY, 401)
    a = np.stack([S.make_clip(50 + i, 1.3) for i in range(3)])
    fb, sb = ctx.fcpe_f0(a, 0.03, return_salience=True)
    for i in range(3):
        f1, s1 = ctx.fcpe_f0(a[i], 0.03, return_salience=True)
        assert np.array_equal(sb[i], s1[0]) and np.array_equal(fb[i], f1[0])


def test_fcpe_checkpoint_file_and_lazy_load(ctx, tmp_path, monkeypatch):
    """fcpe.pt as the reference stores it ({"config", "model"}, FCPE.py:708-736) is opened lazily from FCPE_DIR on
    the first get_f0(f0_method="fcpe") (pipeline.py:169-178); legacy weight_g / weight_v names of dense_out load too."""
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I, pipeline as P
    ck = S.fcpe_checkpoint(S.FCPE_CFG_TINY, 401)
    sd = dict(ck["model"])
    sd["dense_out.weight_g"] = sd.pop("dense_out.parametrizations.weight.original0")
    sd["dense_out.weight_v"] = sd.pop("dense_out.parametrizations.weight.original1")
    ck["model"] = S.to_torch(sd)
    path = tmp_path / "fcpe.pt"
    torch.save(ck, path)
    I._CTX[0] = ctx
    ctx.fcpe_loaded = False
    monkeypatch.setattr(P, "FCPE_DIR", str(path))
    vc = I.VC(48000, I.Config())
    x = np.pad(S.make_clip(40, 1.7), (16000, 16000), mode="reflect").astype(np.float32)
    coarse, f0 = vc.get_f0("x.wav", x, len(x) // 160, 0, "fcpe", 3, 128)
    assert ctx.fcpe_loaded
    _load(ctx, S.FCPE_CFG_TINY, 401)
    c2, f2 = vc.get_f0("x.wav", x, len(x) // 160, 0, "fcpe", 3, 128)
    assert np.array_equal(coarse, c2) and np.array_equal(f0, f2)


def test_pipeline_fcpe_vs_reference_golden(ctx):
    """VC.pipeline(..., f0_method="fcpe", ...) end to end against the reference's output with the same draws."""
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I
    d = np.load(os.path.join(GOLD, "pipeline_tiny_fcpe.npz"))
    hcfg, fcfg, scfg = json.loads(str(d["cfgs"]))
    seed = int(d["seed"])
    I._CTX[0] = ctx
    hub = I.load_hubert("cuda:0", False, None, state=S.hubert_state(hcfg, seed), cfg=hcfg)
    _load(ctx, fcfg, seed)
    ctx.fcpe_loaded = True
    cpt = S.synth_checkpoint(scfg, seed)
    cpt["weight"] = S.synth_state(scfg, seed, input_dim=hcfg["embed_dim"])
    cfg = I.Config()
    cfg.x_pad, cfg.x_query, cfg.x_center, cfg.x_max = [int(v) for v in d["geo"]]
    cpt, version, net_g, tgt_sr, vc = I.get_vc("cuda:0", False, cfg, None, cpt=cpt)
    audio = S.make_clip(int(d["clip"]), float(d["seconds"]))
    noise = np.concatenate([d["z_noise_0"].ravel(), d["src_noise_0"].ravel()]).astype(np.float32)
    pcm, f32 = vc.pipeline(hub, net_g, 0, audio.astype(np.float64), "x.wav", float(d["pitch"]), "fcpe", None, 0, 1, 3,
                           tgt_sr, 0, float(d["volume_envelope"]), "v2", float(d["protect"]), 128, None, noise=noise,
                           return_f32=True)
    ref = d["pcm"]
    assert pcm.shape == ref.shape
    diff = np.abs(pcm.astype(np.int32) - ref.astype(np.int32))
    tp = int(tgt_sr) * int(d["geo"][0])
    e = rms(f32 - d["raw"][tp:-tp])
    print(f"pipeline fcpe: pcm max diff {diff.max()} LSB, frac>1 {np.mean(diff > 1):.2e}; float rms err {e:.3e}")
    assert diff.max() <= 8 and np.mean(diff > 1) < 0.02
    assert e < 1e-4                                              # north-star budget 1e-3


def test_c2_fcpe_full_size_vs_reference_golden(ctx):
    """BASELINE config C2 (30 s, 48 k, full-size models) with f0_method="fcpe" against the reference's own
    VC.pipeline output (every 997-th sample + per-4096-block RMS, tools/gen_golden.py --full).
    Tolerance: the north star's 1e-3 RMS on the float waveform, and why not tighter: fcpe's post-processing
    bridges every unvoiced frame, so the NSF sine source is voiced for all 32 s and integrates f0 without a reset;
    a 1e-6 relative f0 difference per frame (fp32 salience -> 9-bin weighted cents) random-walks into ~1e-3 rad of
    phase by the end of the clip.  The CPU oracle itself sits at 4.6e-5 RMS / 7 LSB from the reference here
    (rmvpe+ C2: 6e-6).  f0 is compared directly: <= 1e-5 relative, coarse identical on >= 99.9 % of frames."""
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I
    from test_gpu_pipeline import _check_blocks
    d = np.load(os.path.join(GOLD, "pipeline_c2_30s_48k_fcpe.npz"))
    hcfg, fcfg, scfg = json.loads(str(d["cfgs"]))
    seed = int(d["seed"])
    I._CTX[0] = ctx
    hub = I.load_hubert("cuda:0", False, None, state=S.hubert_state(hcfg, seed), cfg=hcfg)
    _load(ctx, fcfg, seed)
    ctx.fcpe_loaded = True
    cpt = S.synth_checkpoint(scfg, seed)
    cpt["weight"] = S.synth_state(scfg, seed, input_dim=hcfg["embed_dim"])
    cpt, version, net_g, tgt_sr, vc = I.get_vc("cuda:0", False, I.Config(), None, cpt=cpt)
    audio = S.make_clip(int(d["clip"]), float(d["seconds"]))
    T = int(d["chunk_lens"][0]) // (tgt_sr // 100)
    gen = torch.Generator().manual_seed(int(d["noise_seed"]))
    z = torch.randn((1, scfg[2], T), generator=gen)
    src = torch.randn((1, T * (tgt_sr // 100), 1), generator=gen)
    noise = np.concatenate([z.numpy().ravel(), src.numpy().ravel()])
    pcm, f32 = vc.pipeline(hub, net_g, 0, audio, "x.wav", 0.0, "fcpe", None, 0, 1, 3, tgt_sr, 0, 1.0, "v2", 0.33,
                           128, None, 50, 1100, noise=noise, return_f32=True)
    assert len(pcm) == int(d["chunk_lens"][0]) - 2 * tgt_sr
    diff = np.abs(pcm[::997].astype(np.int32) - d["pcm_samples"].astype(np.int32))
    idx = np.arange(0, int(d["chunk_lens"][0]), 997)
    keep = (idx >= tgt_sr) & (idx < int(d["chunk_lens"][0]) - tgt_sr)
    e = rms(f32[idx[keep] - tgt_sr] - d["raw_samples"][keep])
    x = np.pad(ctx.highpass(audio.astype(np.float64)), (vc.t_pad, vc.t_pad), mode="reflect")
    coarse, f0 = vc.get_f0("x", x, len(d["f0"]), 0.0, "fcpe", 3, 128, None, 50, 1100)
    ef = (np.abs(f0 - d["f0"]) / d["f0"]).max()
    print(f"c2 fcpe: float rms err {e:.3e} (rms {rms(d['raw_samples']):.3f}); pcm max diff {diff.max()} LSB, "
          f"frac>1 {np.mean(diff > 1):.2e}; f0 max rel err {ef:.2e}; coarse differs at "
          f"{int((coarse != d['coarse']).sum())} of {len(coarse)} frames; stage ms {ctx.last_timing()}")
    assert e < 1e-3
    assert diff.max() <= 64
    assert ef < 1e-5 and np.mean(coarse != d["coarse"]) < 1e-3
    _check_blocks(f32, d, tgt_sr, tol=2e-3)


def test_ragged_batch_with_fcpe_equals_single_runs(ctx):
    """f0_method="fcpe" through the batched path (rvcx_convert_batch): two clips of equal length form a micro-batch
    (B = 2 through FCPE, HuBERT and the synthesizer), one clip is cut into chunks by the (1, 1, 2, 3) geometry; each
    utterance is bit-identical to its single run, and the F0 back-end really is FCPE (rmvpe+ gives another result)."""
    from polgen_rvc_amd import _lib, synthetic as S, weights as W
    hcfg, scfg = S.HUBERT_CFG_TINY, S.SYNTH_CFG_TINY
    ctx.load_hubert(W.hubert_cfg_struct(hcfg), S.hubert_state(hcfg, 7))
    ctx.load_rmvpe(W.rmvpe_cfg_struct(S.RMVPE_CFG_TINY), S.rmvpe_state(S.RMVPE_CFG_TINY, 7))
    _load(ctx, S.FCPE_CFG_TINY, 401)
    mid = ctx.load_synth(W.synth_cfg_struct(scfg, hcfg["embed_dim"]), S.synth_state(scfg, 7, input_dim=hcfg["embed_dim"]))

    def params(seed, method):
        p = _lib.Params(2.0, 50.0, 1100.0, 0.0, 0.33, 0.5, 0, 1, 1, 2, 3, seed)
        p.f0_method = method
        return p
    clips = [S.make_clip(60, 1.9), S.make_clip(61, 5.3), S.make_clip(62, 1.9)]
    batch = ctx.convert_batch(mid, clips, params(5, _lib.F0_FCPE))
    for i, c in enumerate(clips):
        alone = ctx.convert_batch(mid, [c], params(5 + i, _lib.F0_FCPE))[0]
        assert np.array_equal(alone, batch[i]), i
    other = ctx.convert_batch(mid, [clips[0]], params(5, _lib.F0_RMVPE))[0]
    assert len(other) == len(batch[0]) and not np.array_equal(other, batch[0])
    _lib.lib().rvcx_unload_synth(ctx._h, mid)


def test_f0_file_branch_of_get_f0_with_fcpe(ctx):
    """pipeline.py:185-191 after the fcpe branch: the (time s, Hz) rows replace the estimate from x_pad seconds on,
    everything else is compute_f0's interpolated track shifted by the pitch."""
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I
    _load(ctx, S.FCPE_CFG_TINY, 401)
    ctx.fcpe_loaded = True
    I._CTX[0] = ctx
    vc = I.VC(4800, I.Config())
    x = np.pad(S.make_clip(5, 1.5).astype(np.float64), (16000, 16000), mode="reflect")
    p_len = len(x) // 160
    inp = np.stack([np.linspace(0.0, 1.0, 11), np.full(11, 200.0)], axis=1)
    base_c, base_f = vc.get_f0("x", x, p_len, 3.0, "fcpe", 3, 128, None)
    coarse, f0 = vc.get_f0("x", x, p_len, 3.0, "fcpe", 3, 128, inp)
    assert coarse.shape == f0.shape == (p_len,)
    assert np.allclose(f0[100:201], 200.0) and (coarse[100:201] == coarse[100]).all() and coarse[100] > 1
    keep = np.r_[0:100, 201:p_len]
    assert np.allclose(f0[keep], base_f[keep], rtol=1e-6) and (coarse[keep] == base_c[keep]).all()


def test_fcpe_region_travels_with_the_weight_broadcast(ctx):
    """The FCPE weights live in their own region, listed after RMVPE's (rvcx_weights_regions): a context that
    loaded zeros of the same shapes reports the same layout and, after the device-side stand-in for the RCCL
    broadcast (rvcx_weights_clone), produces the same F0."""
    from polgen_rvc_amd import _lib, synthetic as S, weights as W
    sd = S.fcpe_state(S.FCPE_CFG_TINY, 401)
    cfg = W.fcpe_cfg_struct(W.fcpe_cfg_from_state(sd))
    src, dst = _lib.Context(0), _lib.Context(0)
    try:
        src.load_fcpe(cfg, sd)
        dst.load_fcpe(cfg, {k: np.zeros_like(v) for k, v in sd.items()})
        ra, ha = src.weights_regions()
        rb, hb = dst.weights_regions()
        assert ha == hb and [n for _, n in ra] == [n for _, n in rb] and len(ra) >= 1
        dst.weights_clone(src)
        x = np.pad(S.make_clip(40, 1.7), (16000, 16000), mode="reflect").astype(np.float32)
        a, sa = src.fcpe_f0(x, 0.03, return_salience=True)
        b, sb = dst.fcpe_f0(x, 0.03, return_salience=True)
        assert np.array_equal(sa, sb) and np.array_equal(a, b) and (a > 0).any()
    finally:
        src.close()
        dst.close()


def test_fcpe_rejects_clips_shorter_than_one_window(ctx):
    """The reference zero-pads such clips (FCPE.py:125-129) -- VC.pipeline never produces them (x_pad seconds of
    padding on both sides) -- and rvcx says so instead of computing something else."""
    from polgen_rvc_amd import _lib, synthetic as S
    _load(ctx, S.FCPE_CFG_TINY, 401)
    with pytest.raises(_lib.RvcxError, match="shorter than one analysis window"):
        ctx.fcpe_f0(np.zeros(800, np.float32), 0.03)
    # the context stays usable after the error
    f0 = ctx.fcpe_f0(S.make_clip(1, 1.0), 0.03)
    assert f0.shape == (1, 101) and np.isfinite(f0).all()
