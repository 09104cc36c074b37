import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", ".."))
from oracle.rmvpe import mel_filterbank
from oracle.fcpe import mel_filterbank as mel_filterbank_slaney_scale


def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm="slaney"):
    assert norm == "slaney"
    fmax = fmax if fmax is not None else sr / 2.0
    if htk:                                                   # RMVPE.py:395-402
        return mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    return mel_filterbank_slaney_scale(sr, n_fft, n_mels, fmin, fmax)   # FCPE.py:115-117 (librosa defaults)
