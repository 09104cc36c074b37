"""Length-bucketed (ragged) micro-batches of rvcx_convert_batch -- SURVEY.md 7 step 9 ("pad-to-bucket by length"),
BASELINE configs[4] (mixed-length TTS utterances; batch conversion is /root/reference/TODO.md:11).

Utterances of one length class share a micro-batch: every network up to the decoder is launched with the class's
geometry and per-item length arrays carry each utterance's own arithmetic.  The contract: a batch item is
bit-identical to its single run (reference semantics: every chunk / utterance is converted independently,
rvc/infer/pipeline.py:381-447; the masks are encoders.py:120-123's x_mask)."""
import numpy as np
import pytest

from conftest import rms

pytestmark = pytest.mark.gpu


def _load(ctx, seed, scfg=None):
    from polgen_rvc_amd import synthetic as S, weights as W
    hcfg, rcfg = S.HUBERT_CFG_TINY, S.RMVPE_CFG_TINY
    scfg = scfg or S.SYNTH_CFG_TINY
    ctx.load_hubert(W.hubert_cfg_struct(hcfg), S.hubert_state(hcfg, seed))
    ctx.load_rmvpe(W.rmvpe_cfg_struct(rcfg), S.rmvpe_state(rcfg, seed))
    st = S.synth_state(scfg, seed, input_dim=hcfg["embed_dim"])
    return ctx.load_synth(W.synth_cfg_struct(scfg, hcfg["embed_dim"]), st)


def _params(index_rate=0.0, protect=0.33, seed=5, volume_envelope=1.0, geo=(1, 6, 38, 41)):
    from polgen_rvc_amd import _lib
    return _lib.Params(0.0, 50.0, 1100.0, index_rate, protect, volume_envelope, 0, *geo, seed)


def _clip(seed, samples):
    from polgen_rvc_amd import synthetic as S
    return S.make_clip(seed, samples / 16000.0 + 1e-4)[:samples].copy()


def test_length_classes_partition_the_lengths(ctx):
    """bucket_length: >= n, idempotent, monotone, and the padded frame counts of a class span bucket_frames frames."""
    mid = _load(ctx, 2)
    p = _params()
    prev = 0
    seen = set()
    for n in range(16160, 16000 * 12, 1777):
        nd = ctx.bucket_length(mid, n, p)
        assert nd >= n and ctx.bucket_length(mid, nd, p) == nd and nd >= prev
        assert (nd + 32000 + 1) % (160 * 32) == 0          # the class ends one sample short of a 32-frame boundary
        assert nd - n < 160 * 128
        prev = nd
        seen.add(nd)
    assert len(seen) >= 6
    # a clip long enough to be cut (x_max = 3 s here) keeps its own geometry
    assert ctx.bucket_length(mid, 16000 * 5, _params(geo=(1, 1, 2, 3))) == 16000 * 5


def test_ragged_micro_batch_equals_single_runs(ctx):
    """Nine clips of nine lengths in four length classes: the call forms four micro-batches, and every clip's PCM and
    float waveform equal, bit for bit, what the clip gives alone (Philox noise, protect, RMS envelope)."""
    mid = _load(ctx, 3)
    p = _params(volume_envelope=0.25)
    lens = [27200, 27360, 27999, 30001, 33333, 36160, 50000, 52000, 71000]
    clips = [_clip(100 + i, n) for i, n in enumerate(lens)]
    classes = {}
    for n in lens:
        classes.setdefault(ctx.bucket_length(mid, n, p), []).append(n)
    assert len(classes) == 4 and max(len(v) for v in classes.values()) >= 3
    pcm, f32 = ctx.convert_batch(mid, clips, p, want_f32=True)
    assert sorted(ctx.last_micro_batches()) == sorted(len(v) for v in classes.values())
    for i, c in enumerate(clips):
        a_pcm, a_f32 = ctx.convert_batch(mid, [c], _params(volume_envelope=0.25, seed=5 + i), want_f32=True)
        assert ctx.last_micro_batches() == [1]
        assert len(a_pcm[0]) == len(pcm[i]) and abs(len(pcm[i]) - (lens[i] // 160) * ctx.synth_upp(mid)) <= 4 * ctx.synth_upp(mid)
        assert np.array_equal(a_f32[0], f32[i]), i
        assert np.array_equal(a_pcm[0], pcm[i]), i
        assert np.isfinite(f32[i]).all() and rms(f32[i]) > 1e-4


def test_ragged_micro_batch_with_index_and_parity_noise(ctx):
    """The same with the retrieval blend on (a shorter member is searched as the matrix of its single run) and with
    caller-supplied parity noise (packed at each member's own frame count)."""
    from polgen_rvc_amd import synthetic as S
    mid = _load(ctx, 4)
    ctx.load_index(S.make_index(2048, S.HUBERT_CFG_TINY["embed_dim"], 1))
    try:
        p = _params(index_rate=0.75)
        lens = [20000, 20480, 24000, 24321, 24800]
        clips = [_clip(200 + i, n) for i, n in enumerate(lens)]
        rng = np.random.default_rng(7)
        noises = [rng.standard_normal(ctx.noise_capacity(mid, n, p)).astype(np.float32) for n in lens]
        pcm, f32 = ctx.convert_batch(mid, clips, p, noises=noises, want_f32=True)
        assert max(ctx.last_micro_batches()) >= 2
        for i, c in enumerate(clips):
            a_pcm, a_f32 = ctx.convert_batch(mid, [c], p, noises=[noises[i]], want_f32=True)
            assert np.array_equal(a_f32[0], f32[i]), i
            assert np.array_equal(a_pcm[0], pcm[i]), i
        # and the blend is live
        q = ctx.convert_batch(mid, [clips[0]], _params(index_rate=0.0), noises=[noises[0]])[0]
        assert not np.array_equal(q, pcm[0])
    finally:
        ctx.load_index(None)


def test_ragged_members_match_the_oracle(ctx):
    """Parity of the masked arithmetic itself: three members of one class against oracle/pipeline.py run on each clip
    alone (same noise): float waveform within 1e-4 RMS, PCM within 8 LSB -- the tolerance of the tiny golden tests."""
    from polgen_rvc_amd import synthetic as S
    from oracle import pipeline as OP
    seed = 6
    hcfg, rcfg, scfg = S.HUBERT_CFG_TINY, S.RMVPE_CFG_TINY, S.SYNTH_CFG_TINY
    mid = _load(ctx, seed)
    models = OP.Models(S.to_torch(S.hubert_state(hcfg, seed)), hcfg, S.to_torch(S.rmvpe_state(rcfg, seed)), rcfg,
                       S.to_torch(S.synth_state(scfg, seed, input_dim=hcfg["embed_dim"])), scfg)
    p = _params()
    lens = [24000, 24480, 26001]
    assert len({ctx.bucket_length(mid, n, p) for n in lens}) == 1
    clips = [_clip(300 + i, n) for i, n in enumerate(lens)]
    want, noises = [], []
    for c in clips:
        opcm, parts = OP.pipeline(models, OP.Geometry(scfg[-1]), c, 0.0, 0, None, 0.0, 1.0, 0.33, 50, 1100, seed=3,
                                  return_parts=True)
        want.append((opcm, parts["audio_f32"]))
        noises.append(np.concatenate([np.concatenate([z.numpy().ravel(), s.numpy().ravel()]) for z, s in parts["noises"]]))
    pcm, f32 = ctx.convert_batch(mid, clips, p, noises=noises, want_f32=True)
    assert ctx.last_micro_batches() == [3]
    for i, (opcm, of32) in enumerate(want):
        assert pcm[i].shape == opcm.shape
        assert rms(f32[i] - of32) < 1e-4, (i, rms(f32[i] - of32))
        assert int(np.abs(pcm[i].astype(np.int32) - opcm.astype(np.int32)).max()) <= 8


def test_cut_clips_keep_equal_length_batches(ctx):
    """Clips long enough to be cut into chunks (x_max = 3 s here) batch only with clips of exactly their length, as
    before; mixed with ragged classes in one call everything still equals its single run."""
    mid = _load(ctx, 5)
    geo = (1, 1, 2, 3)
    p = _params(geo=geo)
    lens = [80000, 80000, 84000, 30000, 30500]
    clips = [_clip(400 + i, n) for i, n in enumerate(lens)]
    pcm = ctx.convert_batch(mid, clips, p)
    assert sorted(ctx.last_micro_batches()) == [1, 2, 2]
    for i, c in enumerate(clips):
        alone = ctx.convert_batch(mid, [c], _params(geo=geo, seed=5 + i))[0]
        assert np.array_equal(alone, pcm[i]), i


def test_the_longest_member_of_a_class_takes_both_paths_to_the_same_bits(ctx):
    """A clip whose length IS its class's geometry runs without length arrays when it is alone (nothing to mask) and with
    them when a shorter clip shares its micro-batch: the masks are no-ops for it, the bits must not move.  Also the
    neighbour one sample shorter, and a class of its own one sample longer."""
    mid = _load(ctx, 7)
    p = _params(volume_envelope=0.5)
    top = ctx.bucket_length(mid, 24000, p)
    assert top > 24000 and ctx.bucket_length(mid, top, p) == top and ctx.bucket_length(mid, top + 1, p) > top + 1
    lens = [top, 24000, top - 1, top + 1]
    clips = [_clip(500 + i, n) for i, n in enumerate(lens)]
    pcm, f32 = ctx.convert_batch(mid, clips, p, want_f32=True)
    assert sorted(ctx.last_micro_batches()) == [1, 3]
    for i, c in enumerate(clips):
        a_pcm, a_f32 = ctx.convert_batch(mid, [c], _params(volume_envelope=0.5, seed=5 + i), want_f32=True)
        assert np.array_equal(a_f32[0], f32[i]) and np.array_equal(a_pcm[0], pcm[i]), i
    # the full-length clip alone really took the mask-free path: a batch of two of them is not ragged either
    two = ctx.convert_batch(mid, [clips[0], clips[0]], p)
    assert ctx.last_micro_batches() == [2] and np.array_equal(two[0], pcm[0])


def test_clips_shorter_than_the_reflect_padding(ctx):
    """``np.pad(audio, (t_pad, t_pad), mode="reflect")`` (pipeline.py:348) reflects repeatedly when the clip is shorter than
    the 1 s padding, so the reference converts a 400-sample clip (the golden ``pipeline_tiny_short`` pins 0.4 s against the
    reference itself).  Clips of 400 samples ... 1.2 s in ONE call: every one equals its single run bit for bit and has
    the reference's length.  Below 400 samples the trim ``audio1[t_pad_tgt:-t_pad_tgt]`` (pipeline.py:441-447) leaves nothing
    and the reference dies in ``np.abs(audio_opt).max()``; below 19 scipy's filtfilt raises: refused here with the reason."""
    from polgen_rvc_amd import _lib
    mid = _load(ctx, 6)
    p = _params()
    lens = [400, 480, 1999, 6400, 15999, 16000, 16001, 19200]
    clips = [_clip(200 + i, n) for i, n in enumerate(lens)]
    pcm, f32 = ctx.convert_batch(mid, clips, p, want_f32=True)
    upp = ctx.synth_upp(mid)
    for i, (n, c) in enumerate(zip(lens, clips)):
        a_pcm, a_f32 = ctx.convert_batch(mid, [c], _params(seed=5 + i), want_f32=True)
        # one chunk: min(p_len, 2 * HuBERT frames) frames, 2 * (x_pad * 100) of them trimmed (pipeline.py:253-256,441-447)
        frames = min((n + 32000) // 160, 2 * ((n + 32000 - 400) // 320 + 1))
        assert len(pcm[i]) == (frames - 200) * upp > 0, (n, len(pcm[i]))
        assert np.array_equal(a_f32[0], f32[i]) and np.array_equal(a_pcm[0], pcm[i]), n
        assert np.isfinite(f32[i]).all()
    for n, why in ((399, "padding"), (161, "padding"), (19, "padding"), (18, "18 samples")):
        with pytest.raises(_lib.RvcxError, match=why):
            ctx.convert_batch(mid, [clips[-1][:n]], p)
    assert len(ctx.convert_batch(mid, [clips[-1][:400]], p)[0]) == 2 * upp      # the context is usable after a refusal
