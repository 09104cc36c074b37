"""Audio I/O edges on the GPU: the device resampler behind ``load_audio`` (rvc/lib/my_utils.py:5-16) and
VC.pipeline's ``resample_sr`` branch (rvc/infer/pipeline.py:453-454) against the oracle's numpy restatement of the same
arithmetic (oracle/audio.py: "kaiser_hq", the Kaiser design to soxr_hq's published targets that is the default since
round 6, and resampy's published "kaiser_best"; parity unpinned against libsoxr itself)."""
import numpy as np
import pytest

from conftest import rms

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sr,channels,seconds", [(44100, 2, 1.3), (48000, 2, 1.0), (48000, 1, 0.7), (22050, 1, 0.5),
                                                 (8000, 1, 0.9)])
def test_resample_to_16k_vs_oracle(ctx, sr, channels, seconds):
    """What the scripts really feed rvc_infer: convert_to_stereo leaves the upload at its original rate, so load_audio
    averages the two channels and converts 44.1 / 48 kHz to 16 kHz."""
    from oracle import audio as OA
    g = np.random.Generator(np.random.PCG64(sr + channels))
    n = int(sr * seconds)
    t = np.arange(n) / sr
    a = 0.4 * np.sin(2 * np.pi * 330.0 * t) + 0.05 * g.standard_normal(n)
    audio = a if channels == 1 else np.stack([a, 0.5 * a + 0.02 * g.standard_normal(n)], 1)
    want = OA.load_audio_from_array(audio, sr, 16000)
    got = ctx.resample(audio, sr, 16000)
    assert got.dtype == np.float64 and got.shape == want.shape
    e = np.abs(got - want).max()
    print(f"{sr} Hz x{channels} -> 16 kHz: {len(got)} samples, max abs diff vs oracle {e:.2e}")
    assert e < 1e-12                                  # same table, same taps, same order: only fma / i0 rounding differs
    # the published resampy filter stays selectable (librosa's default before 0.10), in its published arithmetic
    want_kb = OA.resample_kaiser_best(OA.to_mono(audio), sr, 16000)
    assert np.abs(ctx.resample(audio, sr, 16000, kind=1) - want_kb).max() < 1e-12
    # and the result is a faithful band-limited copy: the 330 Hz tone keeps its level within the filter's known gain
    if channels == 1 and sr > 16000:
        ref = 0.4 * np.sin(2 * np.pi * 330.0 * np.arange(len(got)) / 16000)
        gain = np.dot(got[400:-400], ref[400:-400]) / np.dot(ref[400:-400], ref[400:-400])
        assert abs(gain - 1.0) < 0.02


def test_load_audio_file_flow_44k_stereo(ctx, tmp_path):
    """voice_conversion.py:45-51 then my_utils.py:5-16 on files: a mono 44.1 kHz upload -> convert_to_stereo -> load_audio
    (stereo 44.1 kHz PCM_16 -> mono float64 16 kHz) equals the oracle on the same bytes."""
    from scipy.io import wavfile
    from oracle import audio as OA
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import audio as A, infer as I
    I._CTX[0] = ctx
    clip = np.interp(np.arange(int(44100 * 1.5)) / 44100.0, np.arange(24000) / 16000.0, S.make_clip(8, 1.5))
    wavfile.write(tmp_path / "up.wav", 44100, (clip * 32767).astype(np.int16))
    A.convert_to_stereo(str(tmp_path / "up.wav"), str(tmp_path / "st.wav"))
    y = I.load_audio(str(tmp_path / "st.wav"), 16000)
    sr, pcm = wavfile.read(tmp_path / "st.wav")
    want = OA.load_audio_from_array(pcm.astype(np.float64) / 32768.0, sr, 16000)
    assert sr == 44100 and pcm.shape[1] == 2 and y.shape == want.shape
    assert np.abs(y - want).max() < 1e-12


def test_pipeline_resample_sr_vs_oracle(ctx):
    """pipeline.py:453-454: resample_sr >= 16000 and != tgt_sr -> the float32 output is resampled before the peak
    normalisation and the int16 cast.  The un-resampled float waveform goes through the oracle's resampler (float32
    running sum, as the published loop does on a float32 array) and the reference's last four lines."""
    from oracle import audio as OA
    from polgen_rvc_amd import synthetic as S
    from polgen_rvc_amd.infer import infer as I
    cfgs = (S.HUBERT_CFG_TINY, S.RMVPE_CFG_TINY, S.SYNTH_CFG_TINY)
    I._CTX[0] = ctx
    hub = I.load_hubert("cuda:0", False, None, state=S.hubert_state(cfgs[0], 4), cfg=cfgs[0])
    I.load_rmvpe("cuda:0", state=S.rmvpe_state(cfgs[1], 4), cfg=cfgs[1])
    cpt = S.synth_checkpoint(cfgs[2], 4)
    cpt["weight"] = S.synth_state(cfgs[2], 4, input_dim=cfgs[0]["embed_dim"])
    cpt, version, net_g, tgt_sr, vc = I.get_vc("cuda:0", False, I.Config(), None, cpt=cpt)
    vc.seed = 21
    audio = S.make_clip(12, 2.5)
    args = (hub, net_g, 0, audio, "x.wav", 0.0, "rmvpe+", None, 0, 1, 3, tgt_sr)
    tail = (1.0, "v2", 0.33, 128, None, 50, 1100)
    pcm0, f0 = vc.pipeline(*args, 0, *tail, return_f32=True)                   # resample_sr = 0: off
    assert tgt_sr == 4800
    pcm1, f1 = vc.pipeline(*args, 16000, *tail, return_f32=True)               # 4800 -> 16000
    want = OA.resample_kaiser_hq(f0.astype(np.float64), tgt_sr, 16000, dtype=np.float32)
    assert len(pcm1) == len(f1) == len(want) == int(len(f0) * (16000 / tgt_sr))
    e = rms(f1 - want) / rms(want)
    print(f"resample_sr {tgt_sr} -> 16000: rel diff vs oracle {e:.2e}")
    assert e < 1e-5                                                            # float32 running sum vs float64
    amax = np.abs(f1).max() / 0.99
    ref_pcm = (f1 * (32768 / amax if amax > 1 else 32768)).astype(np.int16)
    assert np.array_equal(pcm1, ref_pcm)
    pcm2 = vc.pipeline(*args, 8000, *tail)                                     # below 16 kHz: ignored (pipeline.py:453)
    assert np.array_equal(pcm2, pcm0)
