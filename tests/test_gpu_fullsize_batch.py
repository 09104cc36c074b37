"""BASELINE configs C3 / C5 at full model size (HuBERT-base, RMVPE E2E(4,1,(2,2)), 48 k and 40 k synthesizers):
the micro-batched conversion path against single runs (bit-exact), the retrieval op at 65 536 x 768 against
float64 brute force, weight regions after a simulated broadcast, and two resident voice models interleaved."""
import os

import numpy as np
import pytest

from conftest import FULL_PCM_BAR, FULL_RMS_BAR, rms

pytestmark = pytest.mark.gpu
SEED = 1900          # the C2 fixture's model seed (tests/golden/pipeline_c2_30s_48k.npz)


@pytest.fixture(scope="module")
def full(ctx):
    """Full-size HuBERT + RMVPE + 48 k voice model resident in the session context."""
    from polgen_rvc_amd import synthetic as S, weights as W
    ctx.load_hubert(W.hubert_cfg_struct(S.HUBERT_CFG_BASE), S.hubert_state(S.HUBERT_CFG_BASE, SEED))
    ctx.load_rmvpe(W.rmvpe_cfg_struct(S.RMVPE_CFG_FULL), S.rmvpe_state(S.RMVPE_CFG_FULL, SEED))
    mid = ctx.load_synth(W.synth_cfg_struct(S.SYNTH_CFG_48K, 768), S.synth_state(S.SYNTH_CFG_48K, SEED))
    yield mid
    from polgen_rvc_amd import _lib
    _lib.lib().rvcx_unload_synth(ctx._h, mid)


def _params(index_rate=0.0, seed=5, volume_envelope=1.0):
    from polgen_rvc_amd import _lib
    return _lib.Params(0.0, 50.0, 1100.0, index_rate, 0.33, volume_envelope, 0, 1, 6, 38, 41, seed)


def _padded(ctx, clip):
    return np.pad(ctx.highpass(clip.astype(np.float64)), (16000, 16000), mode="reflect").astype(np.float32)


def test_index_ids_vs_float64_bruteforce_65536x768(ctx, full):
    """C3's retrieval at its real size: 1599 HuBERT frames of a 30 s clip against 65 536 x 768 stored vectors.
    Neighbour ids bit-exact vs the oracle's float64 brute force, blended features <= 1e-5 relative."""
    from oracle import pipeline as OP
    from polgen_rvc_amd import synthetic as S
    feats = ctx.hubert_features(_padded(ctx, S.make_clip(0, 30.0)), 768)[0]
    assert feats.shape == (1599, 768)
    big = S.make_index_from_feats(feats, 65536, 0)
    ctx.load_index(big)
    try:
        out, ids, dist = ctx.index_blend(feats, 0.75)
        ref, rids, rdist = OP.index_blend(feats, big, 0.75)
        print(f"ids: {int((ids != rids).any(1).sum())} of {len(ids)} queries differ; blend rel err "
              f"{rms(out - ref) / rms(ref):.2e}; dist max abs err {np.abs(dist - rdist).max():.2e}")
        assert (ids == rids).all()
        # faiss' flat search forms |q|^2 + |b|^2 - 2 q.b in float32 as well: distances of ~0.02 .. 1.2 between
        # vectors of squared norm ~780 carry ~1e-4 absolute noise, the blend weights (1/d)^2 inherit it
        assert rms(out - ref) / rms(ref) < 2e-4
        assert np.abs(dist - rdist).max() < 1e-2
        # a query far from every planted neighbour still agrees (random rows only): ids exact where the float64
        # gap between the 8th and 9th neighbour exceeds the fp32 distance noise
        g = np.random.Generator(np.random.PCG64(9))
        q = g.standard_normal((64, 768)).astype(np.float32)
        out2, ids2, _ = ctx.index_blend(q, 0.75)
        d2 = ((q.astype(np.float64) ** 2).sum(1)[:, None] - 2.0 * q.astype(np.float64) @ big.astype(np.float64).T
              + (big.astype(np.float64) ** 2).sum(1)[None, :])
        srt = np.sort(d2, axis=1)
        safe = (srt[:, 8] - srt[:, 7]) > 1e-2
        _, rids2, _ = OP.index_blend(q, big, 0.75)
        print(f"random queries: {int(safe.sum())} safe, "
              f"{int((np.sort(ids2[safe], 1) != np.sort(rids2[safe], 1)).any(1).sum())} differ")
        assert safe.sum() >= 32 and (np.sort(ids2[safe], 1) == np.sort(rids2[safe], 1)).all()
    finally:
        ctx.load_index(None)


def test_c3_batch_of_8_equals_single_runs(ctx, full):
    """B = 8 x 30 s, full-size models, index 65 536 x 768, index_rate 0.75: every utterance of the batched call is
    bit-identical to converting it alone (same Philox stream seed + i), and the batch really ran as one."""
    from polgen_rvc_amd import synthetic as S
    clips = [S.make_clip(i, 30.0) for i in range(8)]
    feats = ctx.hubert_features(_padded(ctx, clips[0]), 768)[0]
    ctx.load_index(S.make_index_from_feats(feats, 65536, 0))
    try:
        p = _params(index_rate=0.75)
        assert ctx.micro_batch(full, len(clips[0]), p) >= 2
        pcm, f32 = ctx.convert_batch(full, clips, p, want_f32=True)
        t_batch = ctx.last_timing()["total"]
        assert all(len(x) == 1439040 for x in pcm)
        t_single = 0.0
        for i, c in enumerate(clips):
            a_pcm, a_f32 = ctx.convert_batch(full, [c], _params(index_rate=0.75, seed=5 + i), want_f32=True)
            t_single += ctx.last_timing()["total"]
            assert np.array_equal(a_f32[0], f32[i]), i
            assert np.array_equal(a_pcm[0], pcm[i]), i
        print(f"8 x 30 s: batched {t_batch:.1f} ms, one at a time {t_single:.1f} ms")
        assert np.isfinite(f32[0]).all() and rms(f32[0]) > 1e-3
        # the blend is live
        p0, _ = ctx.convert_batch(full, [clips[0]], _params(index_rate=0.0), want_f32=True)
        assert not np.array_equal(p0[0], pcm[0])
    finally:
        ctx.load_index(None)


def test_ragged_full_size_batch_equals_single_runs(ctx, full):
    """C5's shape of traffic: utterances of different lengths in one call (two share a length and form a
    micro-batch, one is long enough to be cut into chunks): each equals its single run."""
    from polgen_rvc_amd import synthetic as S
    clips = [S.make_clip(20, 4.0), S.make_clip(21, 7.5), S.make_clip(22, 4.0), S.make_clip(23, 43.0)]
    p = _params(volume_envelope=0.5)
    pcm = ctx.convert_batch(full, clips, p)
    for i, c in enumerate(clips):
        alone = ctx.convert_batch(full, [c], _params(volume_envelope=0.5, seed=5 + i))[0]
        assert np.array_equal(alone, pcm[i]), i
    assert len(pcm[3]) > 42 * 48000 - 4800


def test_weight_regions_survive_a_simulated_broadcast(ctx, full):
    """the weight layout must depend on shapes only.  A second context loads ZERO-filled
    tensors of the same configurations (weight-norm folds of zeros are NaN), must report the same chunk sizes
    and layout hash, receives the first context's chunks (rvcx_weights_clone: device-to-device copies stand in for
    the RCCL broadcast on this 1-GPU box), adopts the flags -- and then converts bit-identically."""
    from polgen_rvc_amd import _lib, synthetic as S, weights as W
    other = _lib.Context(0)
    try:
        z = lambda st: {k: np.zeros_like(v) for k, v in st.items()}
        other.load_hubert(W.hubert_cfg_struct(S.HUBERT_CFG_BASE), z(S.hubert_state(S.HUBERT_CFG_BASE, SEED)))
        other.load_rmvpe(W.rmvpe_cfg_struct(S.RMVPE_CFG_FULL), z(S.rmvpe_state(S.RMVPE_CFG_FULL, SEED)))
        mid2 = other.load_synth(W.synth_cfg_struct(S.SYNTH_CFG_48K, 768), z(S.synth_state(S.SYNTH_CFG_48K, SEED)))
        # the session context may hold more voice models than `full`: compare a fresh source context instead
        src = _lib.Context(0)
        src.load_hubert(W.hubert_cfg_struct(S.HUBERT_CFG_BASE), S.hubert_state(S.HUBERT_CFG_BASE, SEED))
        src.load_rmvpe(W.rmvpe_cfg_struct(S.RMVPE_CFG_FULL), S.rmvpe_state(S.RMVPE_CFG_FULL, SEED))
        mid1 = src.load_synth(W.synth_cfg_struct(S.SYNTH_CFG_48K, 768), S.synth_state(S.SYNTH_CFG_48K, SEED))
        ra, ha = src.weights_regions()
        rb, hb = other.weights_regions()
        assert ha == hb and [n for _, n in ra] == [n for _, n in rb] and len(ra) > 3
        other.weights_clone(src)          # chunk-by-chunk device copies + adopt: what the broadcast does per rank
        clip = S.make_clip(3, 6.0)
        a = src.convert_batch(mid1, [clip], _params(seed=2))[0]
        b = other.convert_batch(mid2, [clip], _params(seed=2))[0]
        assert np.array_equal(a, b) and len(a) > 0 and np.abs(a.astype(np.int32)).max() > 100
        src.close()
    finally:
        other.close()


def test_voice_model_regions_are_freed_on_unload(ctx):
    """loading and unloading voice models / indices repeatedly must not exhaust anything."""
    from polgen_rvc_amd import _lib, synthetic as S, weights as W

    def free_bytes():
        return ctx.mem_info()[0]
    st = S.synth_state(S.SYNTH_CFG_48K, 3)
    cfg = W.synth_cfg_struct(S.SYNTH_CFG_48K, 768)
    mid = ctx.load_synth(cfg, st)
    _lib.lib().rvcx_unload_synth(ctx._h, mid)
    base = free_bytes()
    for _ in range(12):                    # 12 x ~230 MB would be visible
        mid = ctx.load_synth(cfg, st)
        _lib.lib().rvcx_unload_synth(ctx._h, mid)
    big = S.make_index(65536, 768, 0)
    for _ in range(4):                     # 4 x ~400 MB
        ctx.load_index(big)
    ctx.load_index(None)
    assert base - free_bytes() < (64 << 20)


def _fixture_noise(cfg, chunk_len, tgt_sr, noise_seed):
    """The two Gaussian draws of a reference run, regenerated from the private generator's seed the fixture
    records (tools/gen_golden.py: z first, then the source noise)."""
    import torch
    T = chunk_len // (tgt_sr // 100)
    gen = torch.Generator().manual_seed(int(noise_seed))
    z = torch.randn((1, cfg[2], T), generator=gen)
    src = torch.randn((1, T * (tgt_sr // 100), 1), generator=gen)
    return np.concatenate([z.numpy().ravel(), src.numpy().ravel()])


def _check_vs_fixture(d, pre, pcm, f32, tgt_sr):
    stride = int(d[pre + "stride"])
    n_raw = int(d[pre + "chunk_lens"][0])
    assert len(pcm) == n_raw - 2 * tgt_sr
    diff = np.abs(pcm[::stride].astype(np.int32) - d[pre + "pcm_samples"].astype(np.int32))
    idx = np.arange(0, n_raw, stride)
    keep = (idx >= tgt_sr) & (idx < n_raw - tgt_sr)
    e = rms(f32[idx[keep] - tgt_sr] - d[pre + "raw_samples"][keep])
    blocks = 0
    for b, ref in enumerate(d[pre + "block_rms"]):            # every sample: per-4096-block RMS of the reference output
        lo, hi = b * 4096, min(n_raw, (b + 1) * 4096)
        if lo >= tgt_sr and hi <= n_raw - tgt_sr:
            assert abs(rms(f32[lo - tgt_sr: hi - tgt_sr]) - float(ref)) <= 2e-4 * max(1.0, float(ref)) + 5e-6, b
            blocks += 1
    return e, int(diff.max()), float(np.mean(diff > 1)), blocks


def test_c5_two_resident_models_interleaved_vs_reference():
    """BASELINE configs[4] in small, against the REFERENCE (fixture pipeline_c5_two_models, produced by the
    reference's own VC.pipeline): a 40 k and a 48 k voice model at full size share one HuBERT and one RMVPE in a
    context; utterances of different lengths are converted alternately, twice -- every result matches the
    reference within 3e-5 RMS (float) / 4 LSB (PCM) and repeats bit for bit."""
    import json
    from polgen_rvc_amd import _lib, synthetic as S, weights as W
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "pipeline_c5_two_models.npz"))
    seed = int(d["seed"])
    hcfg, rcfg = json.loads(str(d["hcfg"])), json.loads(str(d["rcfg"]))
    ctx = _lib.Context(0)          # its own context: this test brings its own HuBERT / RMVPE (fixture seed)
    ctx.load_hubert(W.hubert_cfg_struct(hcfg), S.hubert_state(hcfg, seed))
    ctx.load_rmvpe(W.rmvpe_cfg_struct(rcfg), S.rmvpe_state(rcfg, seed))
    mids, utts = {}, []
    try:
        for u in range(int(d["n_utts"])):
            pre = f"u{u}_"
            scfg = json.loads(str(d[pre + "scfg"]))
            key = (scfg[-1], int(d[pre + "synth_seed"]))
            if key not in mids:
                mids[key] = ctx.load_synth(W.synth_cfg_struct(scfg, 768), S.synth_state(scfg, key[1]))
            utts.append((pre, scfg, mids[key], S.make_clip(int(d[pre + "clip"]), float(d[pre + "seconds"]))))
        assert len(mids) == 2 and {k[0] for k in mids} == {40000, 48000}
        first = {}
        for rnd in range(2):
            for pre, scfg, mid, clip in utts:                   # alternates 40 k / 48 k / 48 k
                tgt = scfg[-1]
                noise = _fixture_noise(scfg, int(d[pre + "chunk_lens"][0]), tgt, d[pre + "noise_seed"])
                pcm, f32 = ctx.convert_batch(mid, [clip], _params(), noises=[noise], want_f32=True)
                if rnd == 0:
                    e, dmax, frac, blocks = _check_vs_fixture(d, pre, pcm[0], f32[0], tgt)
                    print(f"{pre} {tgt} Hz: float rms err {e:.3e}, pcm max diff {dmax} LSB, {blocks} blocks")
                    assert e < FULL_RMS_BAR and dmax <= FULL_PCM_BAR and frac < 0.02 and blocks > 20, \
                        f"{pre}: float rms err {e:.3e} (bar {FULL_RMS_BAR:g}), pcm max diff {dmax} LSB (bar {FULL_PCM_BAR})"
                    first[pre] = pcm[0]
                else:
                    assert np.array_equal(first[pre], pcm[0])
    finally:
        ctx.close()


def test_c3_item_with_retrieval_blend_vs_reference(ctx, full):
    """BASELINE configs[2], one utterance of the batch against the REFERENCE: fixture pipeline_c3_30s_48k_index is
    the reference's own VC.pipeline output with index_rate 0.75 over a 65 536 x 768 index (its blend code,
    pipeline.py:239-250, around an exact-L2 stand-in for faiss' search).  The index is rebuilt here from the
    GPU's HuBERT features with the same planted-neighbour rule (they differ from the reference-side features by
    ~1e-6, far below the neighbour gaps); the same item inside a batch of 4 must give the same bits."""
    import hashlib
    import json
    from oracle import pipeline as OP
    from polgen_rvc_amd import synthetic as S
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "pipeline_c3_30s_48k_index.npz"))
    assert int(d["seed"]) == SEED and int(d["index_rows"]) == 65536
    scfg = json.loads(str(d["cfgs"]))[2]
    clip = S.make_clip(int(d["clip"]), float(d["seconds"]))
    feats = ctx.hubert_features(_padded(ctx, clip), 768)[0]
    big = S.make_index_from_feats(feats, 65536, 0)
    # neighbour ids of the GPU's features in the GPU-side index == the ids the reference run searched
    _, ids, _ = OP.index_blend(feats, big, 0.75)
    assert hashlib.sha256(ids.astype(np.int64).tobytes()).hexdigest() == str(d["ids_sha256"])
    ctx.load_index(big)
    try:
        noise = _fixture_noise(scfg, int(d["chunk_lens"][0]), 48000, d["noise_seed"])
        p = _params(index_rate=float(d["index_rate"]))
        pcm, f32 = ctx.convert_batch(full, [clip], p, noises=[noise], want_f32=True)
        e, dmax, frac, blocks = _check_vs_fixture(d, "", pcm[0], f32[0], 48000)
        print(f"C3 item vs reference: float rms err {e:.3e}, pcm max diff {dmax} LSB, {blocks} blocks checked")
        assert e < FULL_RMS_BAR and dmax <= FULL_PCM_BAR and frac < 0.02 and blocks > 300, \
            f"C3 item: float rms err {e:.3e} (bar {FULL_RMS_BAR:g}), pcm max diff {dmax} LSB (bar {FULL_PCM_BAR})"
        others = [S.make_clip(40 + i, 30.0) for i in range(3)]
        pcm4 = ctx.convert_batch(full, [others[0], clip, others[1], others[2]], p, noises=[None, noise, None, None])
        assert np.array_equal(pcm4[1], pcm[0])
        # without the blend the waveform is a different one: the fixture really exercises the retrieval path
        # (the planted neighbours sit within 0.005 per channel of the queries, so the blend moves the waveform
        # only a little -- but several times more than the distance to the reference)
        pcm0, f0_ = ctx.convert_batch(full, [clip], _params(), noises=[noise], want_f32=True)
        assert rms(f0_[0] - f32[0]) > 3 * e
    finally:
        ctx.load_index(None)


def test_c5_256_mixed_length_utterances_two_models_ragged_batches():
    """BASELINE configs[4] at its stated size on one GPU: the 256 utterances of U(3, 15) s bench.py's c5 workload
    converts (even -> 40 k, odd -> 48 k voice model, one call per model as app code would issue them), PLUS the three
    utterances of the reference fixture pipeline_c5_two_models riding in the same calls with their own noise.
    * the calls really batch: mixed lengths form ragged micro-batches (length classes), mean size >= 4;
    * 16 sampled utterances (8 per model; shortest, longest and members of full micro-batches among them) are
      bit-equal to their single runs;
    * the three fixture utterances, converted inside those 131- / 128-utterance calls, still match the REFERENCE's own
      VC.pipeline output within 3e-5 RMS (float) / 4 LSB (PCM)."""
    import json
    from bench import c5_lengths
    from polgen_rvc_amd import _lib, synthetic as S, weights as W
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "pipeline_c5_two_models.npz"))
    seed = int(d["seed"])
    hcfg, rcfg = json.loads(str(d["hcfg"])), json.loads(str(d["rcfg"]))
    ctx = _lib.Context(0)
    try:
        ctx.load_hubert(W.hubert_cfg_struct(hcfg), S.hubert_state(hcfg, seed))
        ctx.load_rmvpe(W.rmvpe_cfg_struct(rcfg), S.rmvpe_state(rcfg, seed))
        mids, fix = {}, []
        for u in range(int(d["n_utts"])):
            pre = f"u{u}_"
            scfg = json.loads(str(d[pre + "scfg"]))
            key = (scfg[-1], int(d[pre + "synth_seed"]))
            if key not in mids:
                mids[key] = ctx.load_synth(W.synth_cfg_struct(scfg, 768), S.synth_state(scfg, key[1]))
            noise = _fixture_noise(scfg, int(d[pre + "chunk_lens"][0]), scfg[-1], d[pre + "noise_seed"])
            fix.append((pre, scfg[-1], mids[key], S.make_clip(int(d[pre + "clip"]), float(d[pre + "seconds"])), noise))
        by_rate = {k[0]: m for k, m in mids.items()}
        assert set(by_rate) == {40000, 48000}
        lengths = c5_lengths()
        assert len(lengths) == 256 and min(lengths) >= 3 * 16000 and max(lengths) <= 15 * 16000
        p = _params()
        checked = 0
        for rate, parity in ((40000, 0), (48000, 1)):
            mid = by_rate[rate]
            sel = [i for i in range(256) if i % 2 == parity]
            clips = [S.make_clip(5000 + i, lengths[i] / 16000.0) for i in sel]
            noises = [None] * len(clips)
            mine = [f for f in fix if f[1] == rate]
            for f in mine:                      # the reference's utterances ride along, with the reference's noise
                clips.append(f[3])
                noises.append(f[4])
            pcm, f32 = ctx.convert_batch(mid, clips, p, noises=noises, want_f32=True)
            mbs = ctx.last_micro_batches()
            assert sum(mbs) == len(clips)
            print(f"{rate} Hz: {len(clips)} utterances in {len(mbs)} micro-batches (mean {np.mean(mbs):.1f}, max {max(mbs)})")
            assert np.mean(mbs) >= 4.0 and max(mbs) >= 8
            # single runs of 8 sampled utterances: extremes of the length range + evenly spread ones
            order = np.argsort([len(c) for c in clips[:len(sel)]])
            pick = sorted({int(order[0]), int(order[-1])} | {int(order[k]) for k in np.linspace(5, len(sel) - 6, 6).astype(int)})
            assert len(pick) == 8
            for j in pick:                      # utterance j of a call draws from Philox(seed + j)
                alone, alone32 = ctx.convert_batch(mid, [clips[j]], _params(seed=5 + j), want_f32=True)
                assert np.array_equal(alone32[0], f32[j]), (rate, j)
                assert np.array_equal(alone[0], pcm[j]), (rate, j)
                checked += 1
            for k, f in enumerate(mine):
                j = len(sel) + k
                e, dmax, frac, blocks = _check_vs_fixture(d, f[0], pcm[j], f32[j], rate)
                print(f"  fixture {f[0]} inside the call: float rms err {e:.3e}, pcm max diff {dmax} LSB, {blocks} blocks")
                assert e < FULL_RMS_BAR and dmax <= FULL_PCM_BAR and frac < 0.02 and blocks > 20, \
                        f"{pre}: float rms err {e:.3e} (bar {FULL_RMS_BAR:g}), pcm max diff {dmax} LSB (bar {FULL_PCM_BAR})"
        assert checked == 16
        assert ctx.fp32_reruns() == 0
    finally:
        ctx.close()
