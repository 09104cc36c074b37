"""BASELINE configs[2] (C3) at its STATED size -- a batch of 64 x 30 s clips, RVC v2 48k, rmvpe+, index_rate 0.75 over
a resident 65 536 x 768 index -- plus a 24-clip batch without the index whose three full micro-batches cross the
front-set hand-off (two front sets, ev_done / ev_front / ev_hubdone in csrc/pipeline.hip) at full model size, and
BASELINE configs[1] (C2) compared with the CPU oracle on EVERY one of its 1 439 040 output samples.

Bars: a batched call is bit-identical to converting its utterances one by one (Philox seed + position in the call);
item 0 of the 64-clip call against the reference's own VC.pipeline output (tests/golden/pipeline_c3_30s_48k_index.npz):
float waveform <= 3e-5 RMS (north star 1e-3; measured 1e-5), PCM <= 4 LSB; C2 vs oracle: float <= 3e-5 RMS over all
samples, no sample off by more than 1e-3, PCM <= 6 LSB everywhere (< 6 % off by more than 1 LSB, < 0.5 % by more than 2)."""
import json
import os

import numpy as np
import pytest

from conftest import FULL_PCM_BAR, FULL_RMS_BAR, rms
from test_gpu_fullsize_batch import SEED, _check_vs_fixture, _fixture_noise, _padded, _params

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def full(ctx):
    from polgen_rvc_amd import _lib, synthetic as S, weights as W
    ctx.load_hubert(W.hubert_cfg_struct(S.HUBERT_CFG_BASE), S.hubert_state(S.HUBERT_CFG_BASE, SEED))
    ctx.load_rmvpe(W.rmvpe_cfg_struct(S.RMVPE_CFG_FULL), S.rmvpe_state(S.RMVPE_CFG_FULL, SEED))
    mid = ctx.load_synth(W.synth_cfg_struct(S.SYNTH_CFG_48K, 768), S.synth_state(S.SYNTH_CFG_48K, SEED))
    yield mid
    _lib.lib().rvcx_unload_synth(ctx._h, mid)


def test_c3_64x30s_with_index_equals_single_runs_and_reference(ctx, full):
    """C3 exactly as BASELINE.json states it: B = 64 clips of 30 s, index 65 536 x 768, index_rate 0.75.  Eight
    micro-batches of eight; every utterance bit-equal to its single run; item 0 (the fixture's clip, with the
    fixture's Gaussian draws) against the reference."""
    from polgen_rvc_amd import synthetic as S
    d = np.load(os.path.join(GOLD, "pipeline_c3_30s_48k_index.npz"))
    assert int(d["seed"]) == SEED and int(d["index_rows"]) == 65536
    scfg = json.loads(str(d["cfgs"]))[2]
    clip0 = S.make_clip(int(d["clip"]), float(d["seconds"]))
    clips = [clip0] + [S.make_clip(100 + i, 30.0) for i in range(1, 64)]
    feats = ctx.hubert_features(_padded(ctx, clip0), 768)[0]
    ctx.load_index(S.make_index_from_feats(feats, 65536, 0))
    try:
        noise0 = _fixture_noise(scfg, int(d["chunk_lens"][0]), 48000, d["noise_seed"])
        p = _params(index_rate=float(d["index_rate"]))
        reruns0, fallbacks0 = ctx.fp32_reruns(), ctx.gru_fallbacks()
        mb = ctx.micro_batch(full, len(clip0), p)
        assert 2 <= mb <= 16 and 64 // mb >= 4
        pcm, f32 = ctx.convert_batch(full, clips, p, noises=[noise0] + [None] * 63, want_f32=True)
        t_batch = ctx.last_timing()["total"]
        assert len(pcm) == 64 and all(len(x) == 1439040 for x in pcm)
        e, dmax, frac, blocks = _check_vs_fixture(d, "", pcm[0], f32[0], 48000)
        print(f"C3 64 x 30 s: {t_batch:.0f} ms per call = {64 * 30.0 / (t_batch * 1e-3):.0f} x real time; item 0 vs "
              f"reference: float rms err {e:.3e}, pcm max diff {dmax} LSB, {blocks} blocks")
        assert e < FULL_RMS_BAR and dmax <= FULL_PCM_BAR and frac < 0.02 and blocks > 300, \
            f"C3 64 x 30 s item 0: float rms err {e:.3e} (bar {FULL_RMS_BAR:g}), pcm max diff {dmax} LSB (bar {FULL_PCM_BAR})"
        t_single = 0.0
        for i, c in enumerate(clips):
            a_pcm, a_f32 = ctx.convert_batch(full, [c], _params(index_rate=float(d["index_rate"]), seed=5 + i),
                                             noises=[noise0] if i == 0 else None, want_f32=True)
            t_single += ctx.last_timing()["total"]
            assert np.array_equal(a_f32[0], f32[i]), i
            assert np.array_equal(a_pcm[0], pcm[i]), i
        print(f"one at a time: {t_single:.0f} ms")
        assert all(np.isfinite(x).all() and rms(x) > 1e-3 for x in f32)
        assert ctx.fp32_reruns() == reruns0          # the split-fp16 kernels never left their range
        # the BiGRU's plain-store publish inside one XCD is an observed hardware behaviour; if it ever breaks, every
        # cluster spins into its time-out and the call silently falls back to the single-workgroup kernel -- a slowdown, not an
        # error.  64 batched + 64 single C3-shaped conversions with zero fallbacks keep it a test failure instead.
        assert ctx.gru_fallbacks() == fallbacks0
    finally:
        ctx.load_index(None)


def test_24_equal_clips_cross_the_front_set_handoff(ctx, full):
    """24 (micro-batch cap 8) or 40 (cap 16) equal-length clips, no index: three micro-batches, so front set 0 is handed from micro-batch 0 to
    micro-batch 2 while micro-batch 1 is in the synthesizer -- each clip still equals its single run bit for bit,
    and a second identical call repeats the first."""
    from polgen_rvc_amd import synthetic as S
    p = _params(volume_envelope=0.5)
    mb = ctx.micro_batch(full, 12 * 16000, p)
    assert mb in (8, 16)
    clips = [S.make_clip(200 + i, 12.0) for i in range(24 if mb == 8 else 40)]
    pcm = ctx.convert_batch(full, clips, p)
    assert len(ctx.last_micro_batches()) == 3 and sum(ctx.last_micro_batches()) == len(clips)
    again = ctx.convert_batch(full, clips, p)
    for i, c in enumerate(clips):
        alone = ctx.convert_batch(full, [c], _params(volume_envelope=0.5, seed=5 + i))[0]
        assert np.array_equal(alone, pcm[i]), i
        assert np.array_equal(again[i], pcm[i]), i
    assert len({x.tobytes() for x in pcm}) == len(clips)     # different clips gave different results


def test_c2_every_sample_vs_cpu_oracle(ctx):
    """the full-size goldens look at one sample in 997 plus block RMS.  Here all 1 439 040 samples
    of a C2 conversion (30 s, 48 k, rmvpe+, full-size models) are compared with the pinned CPU oracle
    (oracle/pipeline.py): sign / phase errors inside a block cannot hide.  The oracle needs ~10 minutes of host time
    for this clip, so its waveform is a committed fixture (tests/gen_oracle_c2_full.py wrote it); the Gaussian noise is
    redrawn here from the same torch generator, in the reference's order (z, then source, chunk by chunk)."""
    import torch
    from polgen_rvc_amd import _lib, synthetic as S, weights as W
    d = np.load(os.path.join(GOLD, "oracle_c2_30s_48k_all.npz"))
    seed = int(d["model_seed"])                       # 1900: a seed gen_golden.py vetted (no near-tie salience frames)
    hcfg, rcfg, scfg = S.HUBERT_CFG_BASE, S.RMVPE_CFG_FULL, S.SYNTH_CFG_48K
    hs, rs, ss = S.hubert_state(hcfg, seed), S.rmvpe_state(rcfg, seed), S.synth_state(scfg, seed)
    audio = S.make_clip(int(d["clip"]), float(d["seconds"]))
    gen = torch.Generator().manual_seed(int(d["noise_seed"]))
    noise = np.concatenate([torch.randn(int(n), generator=gen).numpy() for n in d["noise_shapes"].ravel()])
    from oracle.pipeline import to_int16
    opcm = to_int16(d["audio_f32"])               # pipeline.py:457-461 on the oracle's float waveform
    c2 = _lib.Context(0)
    try:
        c2.load_hubert(W.hubert_cfg_struct(hcfg), hs)
        c2.load_rmvpe(W.rmvpe_cfg_struct(rcfg), rs)
        mid = c2.load_synth(W.synth_cfg_struct(scfg, 768), ss)
        pcm, f32 = c2.convert_batch(mid, [audio], _params(), noises=[noise], want_f32=True)
    finally:
        c2.close()
    ref = d["audio_f32"]
    assert pcm[0].shape == opcm.shape == (1439040,) and f32[0].shape == ref.shape
    err = f32[0].astype(np.float64) - ref.astype(np.float64)
    e, emax = rms(err), float(np.abs(err).max())
    dp = np.abs(pcm[0].astype(np.int32) - opcm.astype(np.int32))
    # sign / phase agreement, sample by sample: correlation of the two waveforms
    corr = float(np.dot(f32[0].astype(np.float64), ref.astype(np.float64)) / (rms(f32[0]) * rms(ref) * len(ref)))
    print(f"C2 all {len(ref)} samples vs oracle: rms err {e:.3e} (signal rms {rms(ref):.3f}), max abs err {emax:.3e}, "
          f"pcm max diff {int(dp.max())} LSB, frac > 1 LSB {np.mean(dp > 1):.2e}, correlation {corr:.9f}")
    msg = (f"C2 all samples vs oracle: rms err {e:.3e} (bar {FULL_RMS_BAR:g}; north star 1e-3), max abs {emax:.3e}, "
           f"pcm max diff {int(dp.max())} LSB")
    assert e < FULL_RMS_BAR, msg                      # measured 1.65e-5 (oracle and GPU each ~1e-5 from the reference)
    assert emax < 1e-3 and corr > 0.999999, msg       # measured 1.5e-4
    # 1.6e-5 rms of float error is 0.5 LSB at full scale: a few per cent of the truncated samples differ by 2.  The
    # maximum here is over ALL 1 439 040 samples (the fixture tests look at every 997-th): measured 5, bar 6
    assert dp.max() <= FULL_PCM_BAR + 2 and np.mean(dp > 1) < 0.06 and np.mean(dp > 2) < 0.005, msg
