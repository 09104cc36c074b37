"""Oracle: FCPE F0 estimator (mel -> conv stack -> Performer/Conformer encoder -> 360-bin salience -> Hz).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Functional restatement of
  rvc/lib/predictors/FCPE.py:73-160    (STFT.get_mel: manual reflect pad, torch.stft center=False, slaney mel, log)
  rvc/lib/predictors/FCPE.py:170-197   (softmax_kernel: positive random features of Performer attention)
  rvc/lib/predictors/FCPE.py:227-268   (PCmer / _EncoderLayer)
  rvc/lib/predictors/FCPE.py:312-353   (DepthWiseConv1d, ConformerConvModule, linear_attention)
  rvc/lib/predictors/FCPE.py:422-541   (FastAttention.forward, SelfAttention.forward with local_heads = 0)
  rvc/lib/predictors/FCPE.py:551-704   (FCPE.forward, cents_local_decoder, cent_to_f0)
  rvc/lib/predictors/FCPE.py:739-788   (FCPEInfer.__call__, Wav2Mel.extract_mel at the model's own 16 kHz)
  rvc/lib/predictors/FCPE.py:806-877   (FCPEF0Predictor.compute_f0 / post_process)
Pinned against the reference module itself (tools/gen_golden.py imports it; tests/golden/fcpe_*.npz).  Two pieces
of the import are stand-ins because the packages are absent here: librosa.filters.mel (restated below from
librosa's published Slaney formula -- "parity unpinned" against librosa itself) and local_attention (never
instantiated: the model is built with local_heads = 0).
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

SR, N_FFT, WIN, HOP, N_MELS, FMIN, FMAX = 16000, 1024, 1024, 160, 128, 0.0, 8000.0
HEADS, DIM_HEAD = 8, 64                      # SelfAttention defaults (FCPE.py:445-446), not read from the config
F0_MIN, F0_MAX, OUT_DIMS = 32.70, 1975.5, 360
POST_HOP = 512                               # FCPEF0Predictor.hop_length default: VC.get_f0 does not pass it


def hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, logstep = 1000.0, math.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr=SR, n_fft=N_FFT, n_mels=N_MELS, fmin=FMIN, fmax=FMAX) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with its defaults htk=False, norm='slaney' (FCPE.py:115-117)."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz_slaney(np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2))
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


def mel_spectrogram(audio: torch.Tensor) -> torch.Tensor:
    """Wav2Mel.extract_mel at sample_rate == 16000 (no resampling), keyshift 0: audio (B,n) -> (B, n//160 + 1, 128)."""
    n = audio.shape[-1]
    pad_left = (WIN - HOP) // 2
    pad_right = max((WIN - HOP + 1) // 2, WIN - n - pad_left)
    mode = "reflect" if pad_right < n else "constant"
    y = F.pad(audio.unsqueeze(1), (pad_left, pad_right), mode=mode).squeeze(1)
    spec = torch.stft(y, N_FFT, hop_length=HOP, win_length=WIN, window=torch.hann_window(WIN), center=False,
                      normalized=False, onesided=True, return_complex=True)
    spec = torch.sqrt(spec.real.pow(2) + spec.imag.pow(2) + 1e-9)
    mel = torch.log(torch.clamp(torch.matmul(torch.from_numpy(mel_filterbank()), spec), min=1e-5)).transpose(1, 2)
    n_frames = n // HOP + 1
    if n_frames > mel.shape[1]:
        mel = torch.cat((mel, mel[:, -1:, :]), 1)
    if n_frames < mel.shape[1]:
        mel = mel[:, :n_frames, :]
    return mel


def softmax_kernel(data, proj, is_query, eps=1e-4):
    """FCPE.py:170-197.  data (B,H,T,d), proj (m,d) -> (B,H,T,m)."""
    dn = data.shape[-1] ** -0.25
    ratio = proj.shape[0] ** -0.5
    dash = torch.einsum("...id,jd->...ij", dn * data, proj)
    diag = ((data ** 2).sum(-1) / 2.0 * dn ** 2).unsqueeze(-1)
    if is_query:
        return ratio * (torch.exp(dash - diag - dash.max(dim=-1, keepdim=True).values) + eps)
    return ratio * torch.exp(dash - diag + eps)


def linear_attention(q, k, v):
    """FCPE.py:340-352."""
    k_cumsum = k.sum(dim=-2)
    d_inv = 1.0 / (torch.einsum("...nd,...d->...n", q, k_cumsum) + 1e-8)
    context = torch.einsum("...nd,...ne->...de", k, v)
    return torch.einsum("...de,...nd,...n->...ne", context, q, d_inv)


def self_attention(sd, p, x):
    """SelfAttention.forward (FCPE.py:503-541) with local_heads = 0, no mask, eval mode.  x (B,T,C)."""
    q = F.linear(x, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(x, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(x, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    B, T, _ = q.shape
    q, k, v = (t.view(B, T, HEADS, DIM_HEAD).transpose(1, 2) for t in (q, k, v))
    proj = sd[p + ".fast_attention.projection_matrix"]
    out = linear_attention(softmax_kernel(q, proj, True), softmax_kernel(k, proj, False), v)
    out = out.transpose(1, 2).reshape(B, T, HEADS * DIM_HEAD)
    return F.linear(out, sd[p + ".to_out.weight"], sd[p + ".to_out.bias"])


def conformer_conv(sd, p, x):
    """ConformerConvModule.forward (FCPE.py:322-337): LN -> 1x1 (C -> 4C) -> GLU -> depth-wise k=31 -> Swish -> 1x1."""
    C = x.shape[-1]
    h = F.layer_norm(x, (C,), sd[p + ".net.0.weight"], sd[p + ".net.0.bias"]).transpose(1, 2)
    h = F.conv1d(h, sd[p + ".net.2.weight"], sd[p + ".net.2.bias"])
    a, g = h.chunk(2, dim=1)
    h = a * torch.sigmoid(g)
    w = sd[p + ".net.4.conv.weight"]
    k = w.shape[-1]
    h = F.pad(h, (k // 2, k // 2 - (k + 1) % 2))                       # calc_same_padding, FCPE.py:271-273
    h = F.conv1d(h, w, sd[p + ".net.4.conv.bias"], groups=w.shape[0])
    h = h * torch.sigmoid(h)
    return F.conv1d(h, sd[p + ".net.6.weight"], sd[p + ".net.6.bias"]).transpose(1, 2)


def dense_out_weight(sd):
    """weight_norm(nn.Linear) in either container layout: w = g * v / ||v|| per output row."""
    if "dense_out.parametrizations.weight.original0" in sd:
        g, v = sd["dense_out.parametrizations.weight.original0"], sd["dense_out.parametrizations.weight.original1"]
    else:
        g, v = sd["dense_out.weight_g"], sd["dense_out.weight_v"]
    return v * (g / v.norm(dim=1, keepdim=True))


def n_layers_of(sd) -> int:
    return 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("decoder._layers."))


@torch.no_grad()
def salience(sd: Dict[str, torch.Tensor], mel: torch.Tensor) -> torch.Tensor:
    """FCPE.forward up to the sigmoid (FCPE.py:638-646).  mel (B,T,128) -> (B,T,360)."""
    x = mel.transpose(1, 2)
    x = F.conv1d(x, sd["stack.0.weight"], sd["stack.0.bias"], padding=1)
    x = F.leaky_relu(F.group_norm(x, 4, sd["stack.1.weight"], sd["stack.1.bias"]))
    x = F.conv1d(x, sd["stack.3.weight"], sd["stack.3.bias"], padding=1).transpose(1, 2)
    C = x.shape[-1]
    for i in range(n_layers_of(sd)):
        p = f"decoder._layers.{i}"
        x = x + self_attention(sd, p + ".attn", F.layer_norm(x, (C,), sd[p + ".norm.weight"], sd[p + ".norm.bias"]))
        x = x + conformer_conv(sd, p + ".conformer", x)
    x = F.layer_norm(x, (C,), sd["norm.weight"], sd["norm.bias"])
    return torch.sigmoid(F.linear(x, dense_out_weight(sd), sd["dense_out.bias"]))


def cent_table() -> torch.Tensor:
    """FCPE.py:594-601: linspace in float64 (numpy) of the float32 end points, stored as float32."""
    lo = 1200.0 * torch.log2(torch.Tensor([F0_MIN]) / 10.0)
    hi = 1200.0 * torch.log2(torch.Tensor([F0_MAX]) / 10.0)
    return torch.Tensor(np.linspace(lo[0], hi[0], OUT_DIMS))


def local_decode_hz(y: torch.Tensor, threshold: float, table: torch.Tensor = None) -> torch.Tensor:
    """cents_local_decoder + cent_to_f0 (FCPE.py:673-693): y (B,T,360) -> Hz (B,T); 0 where max(y) <= threshold.
    `table`: the checkpoint's cent_table buffer (a persistent buffer, so load_state_dict overwrites the computed one)."""
    ci = (cent_table() if table is None else table)[None, None, :].expand(y.shape[0], y.shape[1], -1)
    conf, mx = torch.max(y, dim=-1, keepdim=True)
    idx = torch.clamp(torch.arange(0, 9) + (mx - 4), 0, OUT_DIMS - 1)
    ci_l, y_l = torch.gather(ci, -1, idx), torch.gather(y, -1, idx)
    rtn = torch.sum(ci_l * y_l, dim=-1, keepdim=True) / torch.sum(y_l, dim=-1, keepdim=True)
    mask = torch.ones_like(conf)
    mask[conf <= threshold] = float("-INF")
    return (10.0 * 2 ** (rtn * mask / 1200.0))[..., 0]


@torch.no_grad()
def infer_hz(sd, audio: np.ndarray, threshold: float = 0.05) -> np.ndarray:
    """FCPEInfer.__call__ (FCPE.py:739-745) for one signal: (n,) -> (n//160 + 1,) Hz."""
    mel = mel_spectrogram(torch.from_numpy(np.asarray(audio, np.float32))[None])
    return local_decode_hz(salience(sd, mel), threshold, sd.get("cent_table"))[0].numpy()


def post_process(f0: np.ndarray, pad_to: int) -> np.ndarray:
    """FCPEF0Predictor.post_process()[0] (FCPE.py:841-867): nearest resize to pad_to frames, then the unvoiced
    frames are bridged by linear interpolation between their voiced neighbours (np.interp, float64)."""
    f0 = F.interpolate(torch.from_numpy(f0.astype(np.float32))[None, None], size=pad_to, mode="nearest")[0, 0]
    nz = torch.nonzero(f0).squeeze()
    vals = torch.index_select(f0, 0, nz).numpy()
    time_org = POST_HOP / SR * nz.numpy()
    time_frame = np.arange(pad_to) * POST_HOP / SR
    if vals.shape[0] <= 0:
        return np.zeros(pad_to)
    if vals.shape[0] == 1:
        return np.ones(pad_to) * vals[0]
    return np.interp(time_frame, time_org, vals, left=vals[0], right=vals[-1])


def compute_f0(sd, x: np.ndarray, p_len: int, threshold: float = 0.03) -> np.ndarray:
    """FCPEF0Predictor.compute_f0 (FCPE.py:869-877) as VC.get_f0 calls it (pipeline.py:169-179): float64 (p_len,)."""
    f0 = infer_hz(sd, x, threshold)
    if np.all(f0 == 0):
        return np.zeros(p_len)
    return post_process(f0, p_len)
