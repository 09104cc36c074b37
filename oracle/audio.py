"""TEST INFRASTRUCTURE (CPU oracle) -- audio I/O edges of the path: ``load_audio`` (rvc/lib/my_utils.py:5-16), the
``librosa.resample`` calls at my_utils.py:12-13 and rvc/infer/pipeline.py:453-454, and ``convert_to_stereo``
(rvc/scripts/voice_conversion.py:45-51).

PARITY UNPINNED: ``librosa`` / ``soxr`` / ``resampy`` / ``soundfile`` are third-party packages that are neither vendored
in /root/reference nor installed here (the reference does not even pin a librosa version: requirements.txt lists none).
``librosa.resample``'s default ``res_type`` is "soxr_hq" since librosa 0.10 and was "kaiser_best" before.  libsoxr's
coefficients are not published as a formula; its DESIGN TARGETS are (soxr.c, soxr_quality_spec, HQ = 20-bit precision):
linear phase, pass-band flat to 0.9136 x Nyquist(out) (7.31 kHz at 16 kHz), stop-band from 1.0 x Nyquist(out) at -120.4 dB.

Two restatements, both tested against the product's resampler (csrc/audio.hip) to 1e-12:

* ``resample_kaiser_hq`` (round 6; what the product computes by default): a Kaiser-windowed sinc DESIGNED TO soxr_hq's
  published targets -- cut-off in the middle of the transition band (roll-off 0.9568), beta 12.82 (a 125 dB Kaiser
  window), 96 zero crossings per wing (the Kaiser length for a 0.0864 x Nyquist transition), a 2**11-per-crossing table
  with linear interpolation (table error -141 dB), every tap at its exact table position.  Measured at 44.1 k / 48 k ->
  16 k (tests/test_resample_spec.py): pass-band within +-0.001 dB up to 7.31 kHz, -0.44 dB at 7.5 kHz, alias rejection
  below -127 dB from 8.0 kHz on: INSIDE the published targets.  Against a scipy FIR built to the same targets the C2
  benchmark signal rendered at 44.1 kHz differs by 4.1e-5 relative RMS (-88 dB; 4.5e-7 when band-limited to 6.5 kHz: the
  rest lies in the 7.31 ... 8 kHz transition band,
  where the recipe does not fix the shape); against libsoxr itself the difference cannot be measured here and is bounded
  by the same figures: two filters that both meet the recipe agree to the ripple (1e-4) below 7.31 kHz and to 1e-6 of
  full scale above 8 kHz.
* ``resample_kaiser_best``: resampy's published "kaiser_best" band-limited sinc interpolation (Smith's algorithm as
  published with resampy: num_zeros = 64, 2**9 table samples per zero crossing, roll-off 0.9475937167399596, beta
  14.769656459379492, integer table step, output length int(n * ratio)); selectable in the product
  (RVCX_RESAMPLER=kaiser_best).  Against the targets: a constant gain of +0.035 dB, -0.4 dB at 7.3 kHz, aliases at -55 ... -64
  dB between 8 and 9 kHz; 3.1e-3 relative RMS from the soxr_hq-spec FIR on the same signal.

Parity of the conversion itself is defined, tested and benchmarked on 16 kHz input (BASELINE.json: "synthetic 16 kHz mono
clips"), where no resampler runs.

Only tests/ (and tools/) may import this module."""
from __future__ import annotations

import numpy as np

KAISER_BEST = dict(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492)


def sinc_window(num_zeros: int, precision: int, rolloff: float, beta: float):
    """Right half of the Kaiser-windowed sinc interpolation filter: num_zeros * 2**precision + 1 samples."""
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def resample_kaiser_best(x: np.ndarray, sr_orig: int, sr_new: int) -> np.ndarray:
    """1-D band-limited sinc interpolation (the published resampy algorithm), float64 arithmetic."""
    x = np.asarray(x, dtype=np.float64)
    ratio = float(sr_new) / float(sr_orig)
    n_out = int(x.shape[0] * ratio)
    interp_win, num_table = sinc_window(**KAISER_BEST)
    if ratio < 1:
        interp_win = interp_win * ratio
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    scale = min(1.0, ratio)
    time_increment = 1.0 / ratio
    index_step = int(scale * num_table)
    nwin = interp_win.shape[0]
    n_orig = x.shape[0]
    y = np.zeros(n_out, dtype=np.float64)
    t = np.arange(n_out, dtype=np.float64) * time_increment
    n = t.astype(np.int64)
    # left wing: samples x[n - i], i = 0 .. i_max - 1
    frac = scale * (t - n)
    index_frac = frac * num_table
    offset = index_frac.astype(np.int64)
    eta = index_frac - offset
    i_max = np.minimum(n + 1, (nwin - offset) // index_step)
    for i in range(int(i_max.max()) if n_out else 0):
        live = i < i_max
        idx = np.where(live, offset + i * index_step, 0)
        w = interp_win[idx] + eta * interp_delta[idx]
        y += np.where(live, w * x[np.where(live, n - i, 0)], 0.0)
    # right wing: samples x[n + k + 1], k = 0 .. k_max - 1
    frac = scale - frac
    index_frac = frac * num_table
    offset = index_frac.astype(np.int64)
    eta = index_frac - offset
    k_max = np.minimum(n_orig - n - 1, (nwin - offset) // index_step)
    for k in range(int(k_max.max()) if n_out else 0):
        live = k < k_max
        idx = np.where(live, offset + k * index_step, 0)
        w = interp_win[idx] + eta * interp_delta[idx]
        y += np.where(live, w * x[np.where(live, n + k + 1, 0)], 0.0)
    return y


KAISER_HQ = dict(num_zeros=96, precision=11, rolloff=0.9568, beta=12.82)


def resample_kaiser_hq(x: np.ndarray, sr_orig: int, sr_new: int, dtype=np.float64) -> np.ndarray:
    """The default filter of csrc/audio.hip (header above): taps at exact table positions p = (frac + i scale) 2**11,
    linear interpolation between table samples, left wing then right wing, running sum in ``dtype``."""
    x = np.asarray(x, dtype=np.float64)
    ratio = float(sr_new) / float(sr_orig)
    n_out = int(x.shape[0] * ratio)
    win, table = sinc_window(**KAISER_HQ)
    scale = min(1.0, ratio)
    if ratio < 1:
        win = win * ratio
    # The Kaiser window does not reach zero at its edge (1 / I0(beta) = 2e-5 of the peak, x the sinc there: 1e-8): with taps at
    # exact positions the LAST tap of a wing lies within rounding of the edge for some outputs, and whether it is taken
    # would depend on the last bit of p.  The final table sample is set to zero, so the interpolated weight runs to zero
    # continuously and the tap count no longer matters (-160 dB: far below the design's stop-band).
    win[-1] = 0.0
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    nwin, n_orig = win.shape[0], x.shape[0]
    t = np.arange(n_out, dtype=np.float64) * (1.0 / ratio)
    n = t.astype(np.int64)
    frac = scale * (t - n)
    step = scale * table
    y = np.zeros(n_out, dtype=dtype)
    taps = int((nwin - 1) / step) + 2

    def wing(p0, src, ok):
        nonlocal y
        for i in range(taps):
            p = p0 + float(i) * step
            idx = p.astype(np.int64)
            live = (idx < nwin - 1) & ok(i)
            if not live.any():
                break
            idx = np.where(live, idx, 0)
            w = win[idx] + (p - idx) * delta[idx]
            term = np.where(live, w * x[np.where(live, src(i), 0)], 0.0)
            y = (y.astype(np.float64) + term).astype(dtype)
    wing(frac * table, lambda i: n - i, lambda i: n - i >= 0)
    wing((scale - frac) * table, lambda i: n + i + 1, lambda i: n + i + 1 < n_orig)
    return y


def to_mono(audio: np.ndarray) -> np.ndarray:
    """librosa.to_mono(audio.T) of a (frames, channels) array as soundfile returns it (my_utils.py:10-11)."""
    a = np.asarray(audio, dtype=np.float64)
    return a if a.ndim == 1 else a.mean(axis=1)


def load_audio_from_array(audio: np.ndarray, sr: int, sample_rate: int) -> np.ndarray:
    """my_utils.py:9-16 after ``sf.read``: mono mean -> resample (when the rates differ) -> flatten."""
    a = to_mono(audio)
    if sr != sample_rate:
        a = resample_kaiser_hq(a, sr, sample_rate)
    return a.flatten()


def convert_to_stereo_array(y: np.ndarray) -> np.ndarray:
    """voice_conversion.py:45-51 on the array librosa.load(sr=None, mono=False) returns ((frames,) for mono files,
    (channels, frames) otherwise): mono is doubled; the reference's second branch tests ``y.ndim > 2``, which a
    (channels, frames) array never satisfies, so files with more than two channels keep all of them.  Returns
    (frames, channels) as handed to sf.write."""
    y = np.asarray(y)
    if y.ndim == 1:
        y = np.vstack([y, y])
    elif y.ndim > 2:
        y = y[:2, :]
    return y.T
