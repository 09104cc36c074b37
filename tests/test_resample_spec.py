"""CPU: what the product's resamplers (oracle/audio.py == csrc/audio.hip: "kaiser_hq", the default since round 6, and
resampy's published "kaiser_best") do in the frequency domain, against the PUBLISHED design targets of the reference's
resampler: librosa.resample's default res_type "soxr_hq" (rvc/lib/my_utils.py:11-12).  kaiser_hq is DESIGNED to those
targets and this file asserts that it meets them; kaiser_best's deviation is measured and stated.  soxr itself is not installed anywhere in this image, so the two cannot be
compared sample by sample; soxr's quality recipe is public (soxr.c, soxr_quality_spec): HQ = 20-bit precision ->
stop-band rejection 20 * 6.02 = 120.4 dB from 1.0 x Nyquist(out), pass-band end 1 - 0.05 / TO_3dB(120.4) = 0.9136 x
Nyquist(out), linear phase.  This file measures ours with tones, builds a stand-in that meets soxr_hq's published
targets (Kaiser FIR, scipy) and states the waveform difference on the benchmark clip rendered at 44.1 kHz.  The numbers
asserted here are the ones quoted in oracle/audio.py's header and DESIGN.md."""
import numpy as np
import pytest

import polgen_rvc_amd  # noqa: F401
from oracle import audio as OA
from polgen_rvc_amd import synthetic as S

SOXR_HQ_PASS = 1.0 - 0.05 / ((1.6e-6 * 120.41 - 7.5e-4) * 120.41 + 0.646)      # 0.9136 (soxr.c: TO_3dB)


def _tone_gain(f, sr_in, sr_out=16000, n=40000, fn=None):
    t = np.arange(n) / sr_in
    y = (fn or OA.resample_kaiser_best)(np.sin(2 * np.pi * f * t), sr_in, sr_out)
    m = len(y)
    lo, hi = int(0.2 * m), int(0.8 * m)
    if f < sr_out / 2:
        c = np.exp(-2j * np.pi * f * np.arange(m)[lo:hi] / sr_out)
        return 20 * np.log10(2 * abs(np.mean(y[lo:hi] * c)))
    return 20 * np.log10(np.sqrt(np.mean(y[lo:hi] ** 2)) / np.sqrt(0.5) + 1e-30)


@pytest.mark.parametrize("sr_in", [44100, 48000])
def test_kaiser_best_frequency_response_against_soxr_hq_targets(sr_in):
    assert abs(SOXR_HQ_PASS - 0.9136) < 1e-4
    # resampy steps through its filter table with int(scale * 512) instead of scale * 512: a constant gain of
    # (scale * 512) / int(scale * 512) = +0.036 dB (44.1 k) / +0.034 dB (48 k) over the whole pass-band
    scale = 16000 / sr_in
    g_dc = 20 * np.log10(scale * 512 / int(scale * 512))
    assert 0.030 < g_dc < 0.040
    # pass-band: flat (+-0.02 dB around that gain) up to 7.0 kHz; soxr_hq stays flat to 0.9136 * 8 kHz = 7.31 kHz, where
    # kaiser_best (cut-off 0.9476 * 8 kHz = 7.58 kHz at -6 dB, 64 zero crossings) is already 0.4 dB down
    for f in (100, 1000, 3000, 5000, 6500, 7000):
        assert abs(_tone_gain(f, sr_in) - g_dc) < 0.02, f
    assert -0.5 < _tone_gain(7300, sr_in) < -0.25
    assert -4.5 < _tone_gain(7500, sr_in) < -3.5
    # alias rejection: what lies above the new Nyquist frequency folds back at -55 ... -64 dB between 8 and 9 kHz and below
    # -67 dB from 10 kHz on -- soxr_hq's target is -120 dB from 8 kHz
    for f, bar in ((8010, -54.0), (8300, -59.5), (8700, -62.0), (10000, -66.5), (12000, -71.0), (20000, -78.0)):
        assert _tone_gain(f, sr_in) < bar, (f, _tone_gain(f, sr_in))


@pytest.mark.parametrize("sr_in", [44100, 48000])
def test_kaiser_hq_meets_the_published_soxr_hq_targets(sr_in):
    """The default filter (round 6): pass-band flat to 0.9136 x 8 kHz = 7.31 kHz, stop-band from 8.0 kHz below -120.4 dB
    (soxr.c: soxr_quality_spec, HQ = 20 bits).  Tones through the numpy restatement of what csrc/audio.hip computes
    (tests/test_gpu_audio.py holds the GPU to it within 1e-12)."""
    hq = OA.resample_kaiser_hq
    n = 48000 if sr_in == 48000 else 44100          # whole periods of every probe tone in the analysed stretch
    edge = SOXR_HQ_PASS * 8000.0
    for f in (100, 1000, 3000, 5000, 7000, 7200, 7300, int(edge)):
        g = _tone_gain(f, sr_in, n=n, fn=hq)
        assert abs(g) < 0.003, (f, g)               # measured +-0.001 dB (the tone estimator's own error at 44.1 k: 0.002)
    assert -0.6 < _tone_gain(7500, sr_in, n=n, fn=hq) < -0.3          # inside the transition band: -0.44 dB
    for f in (8000, 8010, 8050, 8200, 8700, 10000, 12000, 15000, 20000):
        g = _tone_gain(f, sr_in, n=n, fn=hq)
        assert g < -120.4, (f, g)                   # measured -127 dB and below


def test_kaiser_hq_against_a_scipy_fir_built_to_the_same_targets():
    """Two filters that meet the same recipe: ours (windowed-sinc table, exact tap positions) and a polyphase FIR from
    scipy's Kaiser design (pass 0.9136, stop 1.0 x 8 kHz, 120.4 dB).  On the C2 benchmark signal rendered at 44.1 kHz they
    differ by 4.1e-5 relative RMS (-88 dB) -- the shape inside the 7.31 ... 8 kHz transition band, which the recipe leaves open, acting
    on the clip's white-noise floor; band-limited to 6.5 kHz they agree to 4.5e-7."""
    from scipy import signal
    up, down = 160, 441
    width = (1.0 - SOXR_HQ_PASS) * 8000.0 / (44100 * up / 2)
    ntaps, beta = signal.kaiserord(120.4, width)
    ntaps |= 1
    h = signal.firwin(ntaps, (SOXR_HQ_PASS + 1.0) / 2 * 8000.0, window=("kaiser", beta), fs=44100 * up) * up
    lead = (-(ntaps // 2)) % down
    hp = np.concatenate([np.zeros(lead), h])

    def both(x):
        ours = OA.resample_kaiser_hq(x, 44100, 16000)
        ref = signal.upfirdn(hp, x, up, down)[(ntaps // 2 + lead) // down:][:len(ours)]
        lo, hi = 2000, len(ref) - 2000
        return np.sqrt(np.mean((ours[lo:hi] - ref[lo:hi]) ** 2)) / np.sqrt(np.mean(ref[lo:hi] ** 2))
    x = S.make_clip(0, 5.0, sr=44100).astype(np.float64)
    rel = both(x)
    sos = signal.butter(12, 6500.0, fs=44100, output="sos")
    rel_bl = both(signal.sosfiltfilt(sos, x))
    print(f"kaiser_hq vs scipy soxr_hq-spec FIR, C2 clip 44.1 k -> 16 k: rel rms {rel:.3e}; band-limited to 6.5 kHz: {rel_bl:.3e}")
    assert rel < 1.5e-4 and rel_bl < 5e-6       # measured 4.1e-5 / 4.5e-7


def test_waveform_difference_against_a_soxr_hq_spec_resampler_on_the_benchmark_clip():
    """The C2 benchmark signal rendered at 44.1 kHz (5 s of it) -> 16 kHz, ours against a stand-in built to soxr_hq's
    published targets (Kaiser FIR: pass 0.9136, stop 1.0 x 8 kHz, 120 dB; polyphase 160 / 441).  Stated bound: the two
    differ by 3.1e-3 relative RMS (-50 dB); with the least-squares gain (1.0027 = +0.023 dB, the table-step truncation)
    divided out 1.6e-3 (-56 dB): the band-edge shape between 7.0 and 8 kHz and the -60 dB aliases, both acting on the
    clip's white-noise floor (0.01 rms over 0 ... 22 kHz)."""
    from scipy import signal
    x = S.make_clip(0, 5.0, sr=44100).astype(np.float64)
    ours = OA.resample_kaiser_best(x, 44100, 16000)
    up, down = 160, 441
    width = (1.0 - SOXR_HQ_PASS) * 8000.0 / (44100 * up / 2)
    ntaps, beta = signal.kaiserord(120.4, width)
    ntaps |= 1
    h = signal.firwin(ntaps, (SOXR_HQ_PASS + 1.0) / 2 * 8000.0, window=("kaiser", beta), fs=44100 * up) * up
    # output sample k of ours sits at input time k * 441 / 160: delay the FIR to a whole number of output periods
    lead = (-(ntaps // 2)) % down
    hp = np.concatenate([np.zeros(lead), h])
    ref = signal.upfirdn(hp, x, up, down)[(ntaps // 2 + lead) // down:][:len(ours)]
    n = len(ref)
    lo, hi = 2000, n - 2000
    sig = np.sqrt(np.mean(ref[lo:hi] ** 2))
    rel = np.sqrt(np.mean((ours[lo:hi] - ref[lo:hi]) ** 2)) / sig
    gain = float(np.dot(ours[lo:hi], ref[lo:hi]) / np.dot(ref[lo:hi], ref[lo:hi]))      # least-squares scalar gain
    rel_g = np.sqrt(np.mean((ours[lo:hi] / gain - ref[lo:hi]) ** 2)) / sig
    print(f"kaiser_best vs soxr_hq-spec stand-in, C2 clip 44.1 k -> 16 k: rel rms {rel:.3e}; fitted gain {gain:.5f} "
          f"({20 * np.log10(gain):+.4f} dB), gain removed {rel_g:.3e}")
    assert 2.5e-3 < rel < 4.0e-3            # measured 3.1e-3
    assert 1.0020 < gain < 1.0042           # measured 1.0027: the table-step truncation (+0.36 % at full pass-band weight)
    assert rel_g < 2.5e-3                   # measured 1.6e-3 (-56 dB): band edge + aliases of the clip's white-noise floor
