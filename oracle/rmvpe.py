"""Oracle: RMVPE F0 estimator (mel -> Deep U-Net -> BiGRU -> salience -> cents -> Hz).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional restatement of
  rvc/lib/predictors/RMVPE.py:35-85    (conv-STFT magnitude)
  rvc/lib/predictors/RMVPE.py:379-439  (MelSpectrogram; librosa.filters.mel htk=True)
  rvc/lib/predictors/RMVPE.py:125-376  (BiGRU, ConvBlockRes, encoder/intermediate/decoder, E2E)
  rvc/lib/predictors/RMVPE.py:461-516  (mel2hidden, decode, to_local_average_cents)
The mel filterbank follows librosa's published formula (librosa is absent here: that
piece is "parity unpinned" against librosa itself, and pinned against the stand-in used
to import the reference).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

N_FFT, HOP, N_MELS, SR, FMIN, FMAX = 1024, 160, 128, 16000, 30.0, 8000.0
CENTS_BASE = 1997.3794084376191


def hz_to_mel_htk(f):
    return 2595.0 * np.log10(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_to_hz_htk(m):
    return 700.0 * (10.0 ** (np.asarray(m, dtype=np.float64) / 2595.0) - 1.0)


def mel_filterbank(sr=SR, n_fft=N_FFT, n_mels=N_MELS, fmin=FMIN, fmax=FMAX) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=True, norm='slaney') -> (n_mels, 1+n_fft/2) f32."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz_htk(np.linspace(hz_to_mel_htk(fmin), hz_to_mel_htk(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


def stft_basis(n_fft=N_FFT) -> np.ndarray:
    """RMVPE.py:46-66 forward_basis: rows [Re(F[0..n/2]) ; Im(F[0..n/2])] * hann, (n_fft+2, n_fft) f32."""
    n = np.arange(n_fft)
    k = np.arange(n_fft // 2 + 1)
    ang = 2.0 * np.pi * np.outer(k, n) / n_fft
    basis = np.vstack([np.cos(ang), -np.sin(ang)]).astype(np.float32)   # FloatTensor cast at :53
    win = (0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft))                 # scipy get_window('hann', fftbins=True)
    return (torch.from_numpy(basis) * torch.from_numpy(win).float()).numpy()


def mel_spectrogram(audio: torch.Tensor, basis=None, melfb=None) -> torch.Tensor:
    """RMVPE.py:412-439 with keyshift=0: audio (B,N) f32 -> log-mel (B,128,1+N//160)."""
    basis = torch.from_numpy(stft_basis()) if basis is None else basis
    melfb = torch.from_numpy(mel_filterbank()) if melfb is None else melfb
    x = F.pad(audio[:, None, :], (N_FFT // 2, N_FFT // 2), mode="reflect")
    ft = F.conv1d(x, basis[:, None, :], stride=HOP)
    cut = N_FFT // 2 + 1
    mag = torch.sqrt(ft[:, :cut] ** 2 + ft[:, cut:] ** 2)
    return torch.log(torch.clamp(torch.matmul(melfb, mag), min=1e-5))


def _bn(sd, p, x, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"].float(), sd[p + ".running_var"].float(),
                        sd[p + ".weight"].float(), sd[p + ".bias"].float(), False, 0.0, eps)


def conv_block_res(sd, p, x):
    # RMVPE.py:140-175
    h = F.relu(_bn(sd, p + ".conv.1", F.conv2d(x, sd[p + ".conv.0.weight"].float(), None, padding=1)))
    h = F.relu(_bn(sd, p + ".conv.4", F.conv2d(h, sd[p + ".conv.3.weight"].float(), None, padding=1)))
    if p + ".shortcut.weight" in sd:
        x = F.conv2d(x, sd[p + ".shortcut.weight"].float(), sd[p + ".shortcut.bias"].float())
    return h + x


def e2e_forward(sd: Dict[str, torch.Tensor], cfg: dict, mel: torch.Tensor) -> torch.Tensor:
    """RMVPE.py:373-376.  mel (B,128,T) with T % 2**en_de_layers == 0 -> salience (B,T,360)."""
    nb, nenc, nint = cfg["n_blocks"], cfg["en_de_layers"], cfg["inter_layers"]
    x = mel.transpose(-1, -2).unsqueeze(1)                      # (B,1,T,128)
    x = _bn(sd, "unet.encoder.bn", x)
    skips = []
    for l in range(nenc):                                       # Encoder.forward :228-234
        for b in range(nb):
            x = conv_block_res(sd, f"unet.encoder.layers.{l}.conv.{b}", x)
        skips.append(x)
        x = F.avg_pool2d(x, 2)
    for l in range(nint):                                       # Intermediate :253-256
        for b in range(nb):
            x = conv_block_res(sd, f"unet.intermediate.layers.{l}.conv.{b}", x)
    for l in range(nenc):                                       # Decoder :304-307
        p = f"unet.decoder.layers.{l}"
        x = F.conv_transpose2d(x, sd[p + ".conv1.0.weight"].float(), None, stride=2,
                               padding=1, output_padding=1)
        x = F.relu(_bn(sd, p + ".conv1.1", x))
        x = torch.cat((x, skips[nenc - 1 - l]), dim=1)
        for b in range(nb):
            x = conv_block_res(sd, f"{p}.conv2.{b}", x)
    x = F.conv2d(x, sd["cnn.weight"].float(), sd["cnn.bias"].float(), padding=1)
    x = x.transpose(1, 2).flatten(-2)                           # (B,T,3*128)
    x = gru_bidir(sd, "fc.0.gru", x)
    x = F.linear(x, sd["fc.1.weight"].float(), sd["fc.1.bias"].float())
    return torch.sigmoid(x)


def gru_bidir(sd, p, x):
    """nn.GRU(384,256,bidirectional,batch_first) forward (RMVPE.py:125-137); gate order r,z,n."""
    H = sd[p + ".weight_hh_l0"].shape[1]
    outs = []
    for sfx, rev in (("", False), ("_reverse", True)):
        wih, whh = sd[f"{p}.weight_ih_l0{sfx}"].float(), sd[f"{p}.weight_hh_l0{sfx}"].float()
        bih, bhh = sd[f"{p}.bias_ih_l0{sfx}"].float(), sd[f"{p}.bias_hh_l0{sfx}"].float()
        gi = F.linear(x, wih, bih)                              # (B,T,3H)
        B, T, _ = gi.shape
        h = torch.zeros(B, H)
        ys = [None] * T
        order = range(T - 1, -1, -1) if rev else range(T)
        for t in order:
            gh = F.linear(h, whh, bhh)
            r = torch.sigmoid(gi[:, t, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
            h = (1 - z) * n + z * h
            ys[t] = h
        outs.append(torch.stack(ys, 1))
    return torch.cat(outs, -1)


@torch.no_grad()
def mel2hidden(sd, cfg, mel):
    # RMVPE.py:461-470 (pad to a multiple of 32 by reflection, crop back)
    n = mel.shape[-1]
    pad = min(32 * ((n - 1) // 32 + 1) - n, n)
    hid = e2e_forward(sd, cfg, F.pad(mel.float(), (0, pad), mode="reflect"))
    return hid[:, :n]


def local_average_cents(sal: np.ndarray, thred: float) -> np.ndarray:
    """RMVPE.py:498-516 (vectorised; same arithmetic: float32 salience * float64 mapping)."""
    cm = np.pad(20 * np.arange(360) + CENTS_BASE, (4, 4))
    center = np.argmax(sal, axis=1)
    salp = np.pad(sal, ((0, 0), (4, 4)))
    idx = center[:, None] + np.arange(9)[None, :]
    win = np.take_along_axis(salp, idx, axis=1)
    cmw = cm[idx]
    divided = np.sum(win * cmw, 1) / np.sum(win, 1)
    divided[np.max(salp, axis=1) <= thred] = 0
    return divided


def decode_f0(hidden: np.ndarray, thred=0.03, f0_min=50, f0_max=1100) -> np.ndarray:
    # RMVPE.py:472-476, 494-496
    cents = local_average_cents(hidden, thred)
    f0 = 10 * (2 ** (cents / 1200))
    f0[f0 == 10] = 0
    f0[(f0 < f0_min) | (f0 > f0_max)] = 0
    return f0


@torch.no_grad()
def infer_f0(sd, cfg, audio: np.ndarray, thred=0.03, f0_min=50, f0_max=1100, return_hidden=False):
    """RMVPE0Predictor.infer_from_audio_with_pitch (RMVPE.py:487-496); audio float array (N,)."""
    a = torch.from_numpy(np.asarray(audio)).float().unsqueeze(0)
    mel = mel_spectrogram(a)
    hid = mel2hidden(sd, cfg, mel).squeeze(0).numpy()
    f0 = decode_f0(hid, thred, f0_min, f0_max)
    return (f0, hid, mel) if return_hidden else f0


def unstable_frames(hid: np.ndarray, thred=0.03, f0_min=50, f0_max=1100, rel=2e-4, cents_tol=0.05) -> np.ndarray:
    """Frames whose decoded f0 could change under a relative perturbation ``rel`` of the salience
    (~20x the fp32 rounding noise): a competing maximum >= 2 bins away, an adjacent-bin tie that moves
    the local average by more than ``cents_tol`` cents, a salience max near the voicing threshold, or an
    f0 near the range gate.  Used by tools/gen_golden.py to pick well-conditioned parity instances:
    decisions taken on exact ties are not reproducible across BLAS builds even for the reference."""
    T = hid.shape[0]
    c = np.argmax(hid, axis=1)
    top = hid[np.arange(T), c]
    bad = np.zeros(T, bool)
    masked = hid.copy()
    for off in (-1, 0, 1):
        idx = np.clip(c + off, 0, 359)
        masked[np.arange(T), idx] = -1
    bad |= masked.max(1) >= top * (1 - rel)
    cm = np.pad(20 * np.arange(360) + CENTS_BASE, (4, 4))
    salp = np.pad(hid, ((0, 0), (4, 4)))

    def cents_at(center):
        idx = center[:, None] + np.arange(9)[None, :]
        w = np.take_along_axis(salp, idx, axis=1).astype(np.float64)
        return (w * cm[idx]).sum(1) / w.sum(1)
    base = cents_at(c)
    for off in (-1, 1):
        cc = np.clip(c + off, 0, 359)
        near = hid[np.arange(T), cc] >= top * (1 - rel)
        bad |= near & (np.abs(cents_at(cc) - base) > cents_tol)
    voiced = top > thred
    bad |= np.abs(top - thred) < thred * rel * 10
    f0 = 10 * 2 ** (base / 1200)
    bad |= voiced & ((np.abs(f0 - f0_min) < f0_min * 1e-4) | (np.abs(f0 - f0_max) < f0_max * 1e-4))
    return np.where(bad)[0]
