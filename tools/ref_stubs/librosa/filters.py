import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", ".."))
from oracle.rmvpe import mel_filterbank


def mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, htk=False, norm="slaney"):
    assert htk and norm == "slaney"
    return mel_filterbank(sr, n_fft, n_mels, fmin, fmax if fmax is not None else sr / 2.0)
