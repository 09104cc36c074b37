"""Audio I/O edges of the path: what ``soundfile.read`` / ``librosa.load`` / ``librosa.resample`` /
``soundfile.write`` do for ``load_audio`` (rvc/lib/my_utils.py:5-16) and ``convert_to_stereo``
(rvc/scripts/voice_conversion.py:45-51).

Decoding: when ``soundfile`` is importable (it is wherever the reference runs) it reads the file, so every container the
reference accepts is accepted; otherwise RIFF/WAVE files (PCM 8/16/24/32, float 32/64) are read with scipy, FLAC files
with the library's own host-side decoder (csrc/flac.hip: the whole format, CRC / MD5 checked) and anything else (mp3,
ogg, ...) raises a clear error naming the missing decoder -- this image has neither soundfile nor ffmpeg.
Encoding: ``rvc_infer`` writes WAV bytes whatever the extension (infer.py:153) -- kept, except that a path ending in
".flac" gets a real FLAC stream (``write_output``; 16-bit, lossless, csrc/flac.hip).
Resampling runs on the GPU (``rvcx_resample_f64``, csrc/audio.hip): a Kaiser-windowed sinc designed to the published
targets of ``librosa.resample``'s default (soxr_hq: flat to 0.9136 x Nyquist, -120 dB from Nyquist; measured +-0.001 dB /
-127 dB); libsoxr's own coefficients are unpublished -- parity unpinned, bounded by those targets (oracle/audio.py);
``RVCX_RESAMPLER=kaiser_best`` selects resampy's published filter (librosa's default before 0.10).
"""
from __future__ import annotations

import numpy as np

_MAGIC = {b"fLaC": "FLAC", b"OggS": "Ogg", b"ID3": "MP3", b"\xff\xfb": "MP3", b"\xff\xf3": "MP3", b"\xff\xf2": "MP3",
          b"FORM": "AIFF", b"caff": "CAF"}


def read_audio(path):
    """``soundfile.read(path)`` -> (float64 array (frames,) or (frames, channels), sample rate)."""
    try:
        import soundfile as sf
    except ImportError:
        sf = None
    if sf is not None:
        return sf.read(path)
    with open(path, "rb") as f:
        head = f.read(12)
    if head[:4] == b"fLaC":
        from .. import _lib
        with open(path, "rb") as f:
            pcm, sr, bits = _lib.flac_decode(f.read())
        return pcm.astype(np.float64) / float(1 << (bits - 1)), sr        # soundfile's float64 normalisation
    if head[:4] != b"RIFF" or head[8:12] != b"WAVE":
        kind = next((v for k, v in _MAGIC.items() if head.startswith(k)), "unknown")
        raise ValueError(f"{path}: {kind} container -- this installation has no decoder for it (the 'soundfile' package "
                         "is not installed); only RIFF/WAVE files can be read without it")
    from scipy.io import wavfile
    sr, a = wavfile.read(path)
    if a.dtype == np.uint8:
        a = (a.astype(np.float64) - 128.0) / 128.0
    elif a.dtype == np.int16:
        a = a.astype(np.float64) / 32768.0
    elif a.dtype == np.int32:            # 24-bit files arrive left-justified in int32
        a = a.astype(np.float64) / 2147483648.0
    else:
        a = a.astype(np.float64)
    return a, int(sr)


def write_wav_pcm16(path, data, sr):
    """``soundfile.write(path, data, sr, format="WAV")``: WAV's default subtype is PCM_16 whatever the dtype."""
    try:
        import soundfile as sf
        sf.write(path, data, sr, format="WAV")
        return
    except ImportError:
        pass
    from scipy.io import wavfile
    a = np.asarray(data)
    if a.dtype.kind == "f":              # libsndfile: lrint(x * 0x7FFF), here with saturation
        a = np.clip(np.rint(a.astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16)
    wavfile.write(path, int(sr), a)


def write_output(path, pcm, sr):
    """What rvc_infer does with ``audio_opt`` (int16): ``sf.write(output_path, audio_opt, tgt_sr, format="WAV")``
    (rvc/infer/infer.py:153) -- WAV bytes whatever the extension, so the UI's "mp3" / "flac" / "m4a" outputs are WAVs in
    disguise.  SURVEY.md 8 f3: a path ending in ".flac" gets a real (lossless, 16-bit) FLAC stream instead; every other
    extension keeps the reference's behaviour byte for byte."""
    a = np.asarray(pcm)
    if str(path).lower().endswith(".flac") and a.dtype == np.int16:
        from .. import _lib
        with open(path, "wb") as f:
            f.write(_lib.flac_encode(a, int(sr)))
        return
    from scipy.io import wavfile
    wavfile.write(path, int(sr), a)


def convert_to_stereo(input_path, output_path):
    """rvc/scripts/voice_conversion.py:45-51, same signature: mono files are doubled to two channels, everything else is
    written back unchanged (the reference's ``y.ndim > 2`` branch never fires on a (channels, frames) array), at the
    ORIGINAL sample rate -- so the file ``rvc_infer`` loads next is rarely 16 kHz and load_audio really resamples."""
    a, sr = read_audio(input_path)       # (frames,) or (frames, channels) == librosa.load(sr=None, mono=False).T
    y = a.T if a.ndim > 1 else a
    if y.ndim == 1:
        y = np.vstack([y, y])
    elif y.ndim > 2:
        y = y[:2, :]
    write_wav_pcm16(output_path, y.T, sr)
