"""`mangio-crepe` (rvc/infer/pipeline.py:86-117, 151-152): torchcrepe's network + Viterbi decoder restated in HIP
(csrc/crepe.hip) against the CPU restatement oracle/crepe.py.  torchcrepe / librosa are not vendored with the reference
and not installed: PARITY UNPINNED -- these tests pin the GPU path to the oracle, not to torchcrepe itself.

Tolerances: sigmoid outputs 1e-4 absolute (fp32 convs of up to 65 536 terms); the Viterbi bins of the SAME probabilities
must be identical (float64 dynamic programme, first-maximum argmax, like numpy); frequencies 1e-6 relative."""
import numpy as np
import pytest

from conftest import rms

pytestmark = pytest.mark.gpu
HOP = 128


def _dither(n, seed=0):
    return np.random.default_rng(seed).triangular(-20, 0, 20, size=n).astype(np.float32)


def _oracle_decode(probs, fmin, fmax, noise, batch):
    import torch
    from oracle import crepe as OC
    bins, pitch = [], []
    for i in range(0, probs.shape[0], batch):
        b, p = OC.decode_batch(torch.from_numpy(np.ascontiguousarray(probs[i:i + batch])), fmin, fmax, noise[i:i + batch])
        bins.append(b)
        pitch.append(p)
    return np.concatenate(bins), np.concatenate(pitch)


@pytest.mark.parametrize("capacity,seconds", [("tiny", 3.0), ("full", 1.3)])
def test_crepe_network_and_decoder_vs_oracle(ctx, capacity, seconds):
    """model.Crepe ("tiny" and "full": torchcrepe's two shipped capacities) on 1 + n/128 frames, ragged last batch."""
    from oracle import crepe as OC
    from polgen_rvc_amd import synthetic as S
    st = S.crepe_state(capacity, 3)
    ctx.load_crepe(st)
    x = S.make_clip(5, seconds)
    F = OC.n_frames(len(x), HOP)
    noise = _dither(F)
    xq = x.astype(np.float32)
    xq = xq / np.quantile(np.abs(xq), 0.999)
    opitch, parts = OC.predict(S.to_torch(st), xq, HOP, 50, 1100, noise, batch_size=2 * HOP, return_parts=True)
    pitch, probs, bins = ctx.crepe_predict(x, HOP, 50, 1100, dither=noise, return_parts=True)
    e = float(np.abs(probs - parts["probs"]).max())
    same = float((bins == parts["bins"]).mean())
    print(f"crepe-{capacity}: {F} frames, sigmoid outputs max abs err {e:.2e}, bins equal to the oracle's {same:.4f}")
    assert probs.shape == (F, 360) and e < 1e-4
    # decoding the GPU's OWN probabilities on the CPU must give the GPU's bins exactly
    ob, op = _oracle_decode(probs, 50, 1100, noise, 2 * HOP)
    assert np.array_equal(ob, bins)
    assert np.abs(op - pitch).max() <= 1e-6 * op.max()
    assert same >= 0.99 and np.abs(opitch - pitch)[bins == parts["bins"]].max() <= 1e-6 * opitch.max()


def test_crepe_decoder_on_moving_and_tied_tracks(ctx):
    """core.postprocess + decode.viterbi on hand-made sigmoid tracks: a peak that glides, jumps by more than the
    12-bin transition band, sits outside [fmin, fmax), frames of exact ties, and a batch length that cuts the track in
    unequal pieces -- bins identical to librosa's algorithm restated in numpy, frequencies to 1e-6."""
    from polgen_rvc_amd import synthetic as S
    ctx.load_crepe(S.crepe_state("tiny", 3))
    rng = np.random.default_rng(7)
    F, batch = 700, 96
    centre = np.concatenate([np.linspace(60, 140, 250), np.full(100, 300.0), np.linspace(300, 20, 200),
                             np.full(150, 345.0)])
    j = np.arange(360)[None, :]
    probs = (0.9 * np.exp(-0.5 * ((j - centre[:, None]) / 3.0) ** 2) + 0.05 * rng.random((F, 360))).astype(np.float32)
    probs[400:420] = 0.5                                  # exact ties: every bin equally likely
    probs[420:430] = 0.0
    noise = _dither(F, 3)
    for fmin, fmax in ((50, 1100), (32.7, 1975.5), (200, 400)):
        pitch, bins = ctx.crepe_decode(probs, batch, fmin, fmax, noise)
        ob, op = _oracle_decode(probs, fmin, fmax, noise, batch)
        assert np.array_equal(bins, ob), (fmin, fmax, np.nonzero(bins != ob)[0][:10])
        assert np.abs(op - pitch).max() <= 1e-6 * op.max()
        assert len(set(bins.tolist())) > 20               # the path really moves


def _tiny(ctx, seed=4):
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I
    cfgs = (S.HUBERT_CFG_TINY, S.RMVPE_CFG_TINY, S.SYNTH_CFG_TINY)
    I._CTX[0] = ctx
    hub = I.load_hubert("cuda:0", False, None, state=S.hubert_state(cfgs[0], seed), cfg=cfgs[0])
    I.load_crepe("cuda:0", state=S.crepe_state("tiny", seed))
    cpt = S.synth_checkpoint(cfgs[2], seed)
    cpt["weight"] = S.synth_state(cfgs[2], seed, input_dim=cfgs[0]["embed_dim"])
    return cfgs, hub, cpt


def test_get_f0_and_pipeline_with_mangio_crepe_vs_oracle(ctx):
    """VC.get_f0(..., "mangio-crepe", hop_length) and VC.pipeline(f0_method="mangio-crepe") through the mirror: quantile
    normalisation, predict, resize to p_len, pitch shift, coarse -- and the whole conversion -- against the oracle with
    the same dither and Gaussian noise.  hop_length 128 (the reference's default) and 64."""
    import torch
    from oracle import crepe as OC, pipeline as OP
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I
    cfgs, hub, cpt = _tiny(ctx)
    sd = S.to_torch(S.crepe_state("tiny", 4))
    cfg = I.Config()
    cpt, version, net_g, tgt_sr, vc = I.get_vc("cuda:0", False, cfg, None, cpt=cpt)
    audio = S.make_clip(33, 2.5)
    for hop in (128, 64):
        xpad = np.pad(OP.highpass(audio.astype(np.float64)), (vc.t_pad, vc.t_pad), mode="reflect")
        p_len = xpad.shape[0] // 160
        dith = _dither(OC.n_frames(xpad.shape[0], hop), hop)
        of0 = OC.get_f0_crepe(sd, xpad, 50, 1100, p_len, hop, dith)
        ocoarse, of0s = OP.f0_to_coarse(of0, 3.0, 50, 1100)
        coarse, f0 = vc.get_f0("x.wav", xpad, p_len, 3.0, "mangio-crepe", 3, hop, crepe_dither=dith)
        assert coarse.shape == (p_len,) and f0.shape == (p_len,)
        bad = int((coarse != ocoarse[:p_len]).sum())
        print(f"hop {hop}: get_f0 f0 max rel err {np.abs(f0 - of0s[:p_len]).max() / of0s.max():.2e}, coarse bins differing {bad}")
        assert np.abs(f0 - of0s[:p_len]).max() <= 2e-6 * of0s.max() and bad == 0
        assert np.allclose(vc.get_f0_crepe(xpad, 50, 1100, p_len, hop, dither=dith), of0, rtol=2e-6, atol=1e-4)
    # the whole conversion (hop 128)
    models = OP.Models(S.to_torch(S.hubert_state(cfgs[0], 4)), cfgs[0], None, None, S.to_torch(cpt["weight"]), cfgs[2])
    models.crepe_sd = sd
    dith = _dither(OC.n_frames(len(audio) + 2 * vc.t_pad, 128), 11)
    opcm, parts = OP.pipeline(models, OP.Geometry(tgt_sr), audio, 2.0, 0, None, 0.0, 1.0, 0.33, 50, 1100, seed=5,
                              return_parts=True, f0_method="mangio-crepe", hop_length=128, crepe_dither=dith)
    noise = np.concatenate([np.concatenate([z.numpy().ravel(), s.numpy().ravel()]) for z, s in parts["noises"]])
    pcm, f32 = vc.pipeline(hub, net_g, 0, audio, "x.wav", 2.0, "mangio-crepe", None, 0, 1, 3, tgt_sr, 0, 1.0, "v2", 0.33,
                           128, None, 50, 1100, noise=noise, return_f32=True, crepe_dither=dith)
    e = rms(f32 - parts["audio_f32"])
    d = int(np.abs(pcm.astype(np.int32) - opcm.astype(np.int32)).max())
    print(f"pipeline with mangio-crepe: float rms err {e:.3e}, pcm max diff {d} LSB")
    assert pcm.shape == opcm.shape and e < 1e-4 and d <= 8
    # without an explicit dither the mirror draws torchcrepe's way: numpy's global generator decides
    np.random.seed(123)
    a = vc.pipeline(hub, net_g, 0, audio, "x.wav", 2.0, "mangio-crepe", None, 0, 1, 3, tgt_sr, 0, 1.0, "v2", 0.33, 128,
                    None, 50, 1100, noise=noise)
    np.random.seed(123)
    b = vc.pipeline(hub, net_g, 0, audio, "x.wav", 2.0, "mangio-crepe", None, 0, 1, 3, tgt_sr, 0, 1.0, "v2", 0.33, 128,
                    None, 50, 1100, noise=noise)
    assert np.array_equal(a, b) and not np.array_equal(a, pcm)


def test_crepe_batch_equals_single_runs(ctx):
    """Utterances of a batch take the crepe path one by one inside their micro-batch: equal to converting them alone."""
    from polgen_rvc_amd import _lib, synthetic as S, weights as W
    seed = 4
    cfgs = (S.HUBERT_CFG_TINY, S.RMVPE_CFG_TINY, S.SYNTH_CFG_TINY)
    ctx.load_hubert(W.hubert_cfg_struct(cfgs[0]), S.hubert_state(cfgs[0], seed))
    ctx.load_crepe(S.crepe_state("tiny", seed))
    mid = ctx.load_synth(W.synth_cfg_struct(cfgs[2], cfgs[0]["embed_dim"]), S.synth_state(cfgs[2], seed, input_dim=cfgs[0]["embed_dim"]))
    p = _lib.Params(1.0, 50.0, 1100.0, 0.0, 0.33, 1.0, 0, 1, 6, 38, 41, 9, _lib.F0_CREPE, 0, 128, 0)
    clips = [S.make_clip(40 + i, s) for i, s in enumerate((2.0, 2.0, 3.1))]
    dith = [_dither(ctx.crepe_frames(len(c) + 32000, 128), i) for i, c in enumerate(clips)]
    together = ctx.convert_batch(mid, clips, p, crepe_dither=dith)
    for i, c in enumerate(clips):
        pi = _lib.Params(1.0, 50.0, 1100.0, 0.0, 0.33, 1.0, 0, 1, 6, 38, 41, 9 + i, _lib.F0_CREPE, 0, 128, 0)
        alone = ctx.convert_batch(mid, [c], pi, crepe_dither=[dith[i]])[0]
        assert np.array_equal(alone, together[i]), i
    _lib.lib().rvcx_unload_synth(ctx._h, mid)


def test_a_silent_clip_gets_an_all_zero_track_and_the_batch_goes_on(ctx):
    """np.quantile(|x|, 0.999) of a silent clip is 0 -- the reference divides by it, every frame turns NaN and
    get_f0_crepe's nan_to_num leaves an all-zero f0 (pipeline.py:91,111-118).  Here the silent item gets exactly that
    (f0 = 0, coarse = 1) instead of failing the call, and the other item of the batch equals its single run."""
    from polgen_rvc_amd import _lib, synthetic as S, weights as W
    seed = 4
    cfgs = (S.HUBERT_CFG_TINY, S.RMVPE_CFG_TINY, S.SYNTH_CFG_TINY)
    ctx.load_hubert(W.hubert_cfg_struct(cfgs[0]), S.hubert_state(cfgs[0], seed))
    ctx.load_crepe(S.crepe_state("tiny", seed))
    mid = ctx.load_synth(W.synth_cfg_struct(cfgs[2], cfgs[0]["embed_dim"]), S.synth_state(cfgs[2], seed, input_dim=cfgs[0]["embed_dim"]))
    try:
        p = _lib.Params(1.0, 50.0, 1100.0, 0.0, 0.33, 1.0, 0, 1, 6, 38, 41, 9, _lib.F0_CREPE, 0, 128, 0)
        voiced, silent = S.make_clip(44, 2.0), np.zeros(32000, np.float32)
        dith = [_dither(ctx.crepe_frames(32000 + 32000, 128), i) for i in range(2)]
        both = ctx.convert_batch(mid, [voiced, silent], p, crepe_dither=dith)
        alone = ctx.convert_batch(mid, [voiced], p, crepe_dither=[dith[0]])[0]
        assert np.array_equal(alone, both[0])
        assert len(both[1]) > 0 and np.isfinite(both[1].astype(np.float64)).all()
        coarse, f0 = ctx.get_f0_crepe_x(np.zeros(64000, np.float32), 400, p, dither=dith[1])
        assert np.all(f0 == 0.0) and np.all(coarse == 1)
    finally:
        _lib.lib().rvcx_unload_synth(ctx._h, mid)
