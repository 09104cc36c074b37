"""Oracle: VC.pipeline / VC.vc / VC.get_f0 host logic around the three networks.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates
  rvc/infer/pipeline.py:19-22   (Butterworth high-pass coefficients)
  rvc/infer/pipeline.py:29-61   (AudioProcessor.change_rms; librosa.feature.rms restated)
  rvc/infer/pipeline.py:132-201 (get_f0 with "rmvpe+": shift, mel-scale coarse quantise)
  rvc/infer/pipeline.py:203-287 (vc: HuBERT -> index blend -> x2 -> protect -> synth)
  rvc/infer/pipeline.py:289-467 (pipeline: filtfilt, chunking, trim, concat, envelope, int16)
FAISS (faiss-cpu==1.7.3, not vendored) is restated as exact brute-force squared-L2 top-8:
"parity unpinned" against faiss itself.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F
from scipy import signal

from . import fcpe as O_fcpe
from . import hubert as O_hubert
from . import rmvpe as O_rmvpe
from . import synth as O_synth

SR = 16000
WINDOW = 160
BH, AH = signal.butter(N=5, Wn=48, btype="high", fs=SR)          # pipeline.py:19-22


class Geometry:
    """VC.__init__ (pipeline.py:66-84) derived sample counts."""

    def __init__(self, tgt_sr, x_pad=1, x_query=6, x_center=38, x_max=41):
        self.tgt_sr = tgt_sr
        self.x_pad, self.x_query, self.x_center, self.x_max = x_pad, x_query, x_center, x_max
        self.t_pad = SR * x_pad
        self.t_pad_tgt = tgt_sr * x_pad
        self.t_pad2 = self.t_pad * 2
        self.t_query = SR * x_query
        self.t_center = SR * x_center
        self.t_max = SR * x_max


def highpass(audio: np.ndarray) -> np.ndarray:
    return signal.filtfilt(BH, AH, audio)                         # pipeline.py:329


def chunk_points(audio: np.ndarray, geo: Geometry):
    """pipeline.py:330-344: quietest sample near every t_center (audio already filtered)."""
    audio_pad = np.pad(audio, (WINDOW // 2, WINDOW // 2), mode="reflect")
    opt_ts = []
    if audio_pad.shape[0] > geo.t_max:
        audio_sum = np.zeros_like(audio)
        for i in range(WINDOW):
            audio_sum += audio_pad[i:i - WINDOW]
        for t in range(geo.t_center, audio.shape[0], geo.t_center):
            seg = np.abs(audio_sum[t - geo.t_query:t + geo.t_query])
            opt_ts.append(t - geo.t_query + np.where(seg == seg.min())[0][0])
    return opt_ts


def f0_to_coarse(f0: np.ndarray, pitch: float, f0_min=50, f0_max=1100, inp_f0=None, x_pad: int = 1):
    """pipeline.py:148-150,183-201.  ``inp_f0``: the (rows, 2) float32 table VC.pipeline parses from an f0 file
    (pipeline.py:349-360), applied exactly as pipeline.py:185-191 does.  Returns (coarse int, f0 Hz)."""
    f0_mel_min = 1127 * np.log(1 + f0_min / 700)
    f0_mel_max = 1127 * np.log(1 + f0_max / 700)
    f0 = f0 * pow(2, pitch / 12)
    tf0 = SR // WINDOW
    if inp_f0 is not None:
        delta_t = np.round((inp_f0[:, 0].max() - inp_f0[:, 0].min()) * tf0 + 1).astype("int16")
        replace_f0 = np.interp(list(range(delta_t)), inp_f0[:, 0] * 100, inp_f0[:, 1])
        shape = f0[x_pad * tf0: x_pad * tf0 + len(replace_f0)].shape[0]
        f0[x_pad * tf0: x_pad * tf0 + len(replace_f0)] = replace_f0[:shape]
    f0bak = f0.copy()
    f0_mel = 1127 * np.log(1 + f0 / 700)
    f0_mel[f0_mel > 0] = (f0_mel[f0_mel > 0] - f0_mel_min) * 254 / (f0_mel_max - f0_mel_min) + 1
    f0_mel[f0_mel <= 1] = 1
    f0_mel[f0_mel > 255] = 255
    return np.rint(f0_mel).astype(int), f0bak


def index_blend(feats: np.ndarray, big_npy: np.ndarray, index_rate: float, k: int = 8):
    """pipeline.py:239-250 with index.search == exact squared-L2 top-k (IndexFlatL2 semantics).
    feats (T,D) f32 -> (blended (T,D) f32, ids (T,k) int64, dist (T,k) f32)."""
    q = feats.astype(np.float64)
    b = big_npy.astype(np.float64)
    d2 = (q * q).sum(1)[:, None] - 2.0 * q @ b.T + (b * b).sum(1)[None, :]
    ix = np.argsort(d2, axis=1, kind="stable")[:, :k]
    score = np.take_along_axis(d2, ix, axis=1).astype(np.float32)
    weight = np.square(1 / score)
    weight /= weight.sum(axis=1, keepdims=True)
    npy = np.sum(big_npy[ix] * np.expand_dims(weight, axis=2), axis=1)
    out = npy * index_rate + (1 - index_rate) * feats
    return out.astype(np.float32), ix.astype(np.int64), score


def index_blend_ivf(feats: np.ndarray, big_npy: np.ndarray, centroids: np.ndarray, assign: np.ndarray,
                    index_rate: float, k: int = 8):
    """pipeline.py:239-250 when ``index`` is a faiss "IVF{nlist},Flat" searched with nprobe = 1 (what RVC index
    files are): the coarse quantiser picks the nearest centroid, only that inverted list is scanned; a list with
    fewer than k vectors pads with id -1 / distance inf, and the reference's ``big_npy[ix]`` then reads the LAST row
    with weight (1/inf)^2 = 0.  float64 distances; parity with faiss itself unpinned."""
    q = feats.astype(np.float64)
    c = centroids.astype(np.float64)
    dc = (q * q).sum(1)[:, None] - 2.0 * q @ c.T + (c * c).sum(1)[None, :]
    qlist = np.argmin(dc, axis=1)
    b = big_npy.astype(np.float64)
    d2 = (q * q).sum(1)[:, None] - 2.0 * q @ b.T + (b * b).sum(1)[None, :]
    d2 = np.where(assign[None, :] == qlist[:, None], d2, np.inf)
    ix = np.argsort(d2, axis=1, kind="stable")[:, :k]
    score = np.take_along_axis(d2, ix, axis=1).astype(np.float32)
    ix = np.where(np.isinf(score), -1, ix)
    with np.errstate(divide="ignore", invalid="ignore"):
        weight = np.square(1 / score)
        weight /= weight.sum(axis=1, keepdims=True)
        npy = np.sum(big_npy[ix] * np.expand_dims(weight, axis=2), axis=1)
        out = npy * index_rate + (1 - index_rate) * feats
    return out.astype(np.float32), ix.astype(np.int64), score


def frame_rms(y: np.ndarray, frame_length: int, hop_length: int) -> np.ndarray:
    """librosa.feature.rms(y, frame_length, hop_length) (center=True, zero pad) -> (1, n_frames)."""
    pad = frame_length // 2
    yp = np.pad(y, (pad, pad), mode="constant")
    n = 1 + (len(yp) - frame_length) // hop_length
    idx = np.arange(frame_length)[None, :] + hop_length * np.arange(n)[:, None]
    power = np.mean(np.abs(yp[idx]) ** 2, axis=1)
    return np.sqrt(power)[None, :]


def change_rms(source_audio, source_rate, target_audio, target_rate, rate):
    # pipeline.py:29-61
    rms1 = frame_rms(source_audio, source_rate // 2 * 2, source_rate // 2)
    rms2 = frame_rms(target_audio, target_rate // 2 * 2, target_rate // 2)
    rms1 = F.interpolate(torch.from_numpy(rms1).float().unsqueeze(0), size=target_audio.shape[0],
                         mode="linear").squeeze()
    rms2 = F.interpolate(torch.from_numpy(rms2).float().unsqueeze(0), size=target_audio.shape[0],
                         mode="linear").squeeze()
    rms2 = torch.maximum(rms2, torch.zeros_like(rms2) + 1e-6)
    return target_audio * (torch.pow(rms1, 1 - rate) * torch.pow(rms2, rate - 1)).numpy()


def to_int16(audio_opt: np.ndarray) -> np.ndarray:
    # pipeline.py:457-461
    audio_max = np.abs(audio_opt).max() / 0.99
    max_int16 = 32768
    if audio_max > 1:
        max_int16 /= audio_max
    return (audio_opt * max_int16).astype(np.int16)


class Models:
    """Bundle of state dicts + configs the oracle needs (all torch tensors, CPU)."""

    def __init__(self, hubert_sd, hubert_cfg, rmvpe_sd, rmvpe_cfg, synth_sd, synth_cfg, fcpe_sd=None, version="v2"):
        self.fcpe_sd = fcpe_sd
        self.version = version          # of the voice model (infer.py:91-97): "v1" = HuBERT layer 9 + final_proj, input_dim 256
        self.hubert_sd, self.hubert_cfg = hubert_sd, hubert_cfg
        self.rmvpe_sd, self.rmvpe_cfg = rmvpe_sd, rmvpe_cfg
        self.synth_sd, self.synth_cfg = synth_sd, synth_cfg


@torch.no_grad()
def vc_chunk(models: Models, audio0: np.ndarray, pitch: np.ndarray, pitchf: np.ndarray, sid: int,
             big_npy: Optional[np.ndarray], index_rate: float, protect: float,
             z_noise: torch.Tensor, src_noise: torch.Tensor, return_parts=False):
    """VC.vc (pipeline.py:203-287), f0 guided.  pitch/pitchf are the per-chunk slices.  ``models.version``: "v2" (output
    layer 12, 768-dim features) or "v1" (output layer 9 + ``final_proj``, 256-dim: pipeline.py:228-236)."""
    feats = torch.from_numpy(audio0).float().view(1, -1)
    if getattr(models, "version", "v2") == "v1":
        feats = O_hubert.final_proj(models.hubert_sd, O_hubert.extract_features(models.hubert_sd, models.hubert_cfg, feats, 9))
    else:
        feats = O_hubert.extract_features(models.hubert_sd, models.hubert_cfg, feats, 12)
    feats0 = feats.clone() if protect < 0.5 else None
    ids = None
    if big_npy is not None and index_rate != 0:
        blended, ids, _ = index_blend(feats[0].numpy(), big_npy, index_rate)
        feats = torch.from_numpy(blended).unsqueeze(0)
    feats = F.interpolate(feats.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)
    if feats0 is not None:
        feats0 = F.interpolate(feats0.permute(0, 2, 1), scale_factor=2).permute(0, 2, 1)
    p_len = audio0.shape[0] // WINDOW
    pitch_t = torch.from_numpy(np.asarray(pitch)).long().unsqueeze(0)
    pitchf_t = torch.from_numpy(np.asarray(pitchf)).float().unsqueeze(0)
    if feats.shape[1] < p_len:
        p_len = feats.shape[1]
        pitch_t, pitchf_t = pitch_t[:, :p_len], pitchf_t[:, :p_len]
    if feats0 is not None:
        pitchff = pitchf_t.clone()
        pitchff[pitchf_t > 0] = 1
        pitchff[pitchf_t < 1] = protect
        pitchff = pitchff.unsqueeze(-1)
        feats = feats * pitchff + feats0 * (1 - pitchff)
    out = O_synth.synthesizer_infer(models.synth_sd, models.synth_cfg, feats, torch.tensor([p_len]),
                                    pitch_t, pitchf_t, torch.tensor([sid]), z_noise, src_noise)
    audio1 = out[0, 0].numpy()
    if return_parts:
        return audio1, dict(feats=feats, ids=ids, p_len=p_len)
    return audio1


def chunk_plan(n_audio: int, opt_ts, geo: Geometry):
    """The (start, end) sample ranges of audio_pad fed to vc() by pipeline.py:381-447, and the
    matching pitch frame ranges."""
    plan, s, t = [], 0, None
    n_pad = n_audio + 2 * geo.t_pad
    for t0 in opt_ts:
        t = t0 // WINDOW * WINDOW
        plan.append((s, t + geo.t_pad2 + WINDOW, s // WINDOW, (t + geo.t_pad2) // WINDOW))
        s = t
    if t is None:
        plan.append((0, n_pad, 0, None))
    else:
        plan.append((t, n_pad, t // WINDOW, None))
    return plan


def noise_shapes(models: Models, n_chunk_samples: int):
    """Shapes of the two Gaussian draws vc() makes for a chunk of n samples."""
    c = O_synth.cfg_fields(models.synth_cfg)
    t_hub = hubert_frames(n_chunk_samples, models.hubert_cfg)
    T = min(n_chunk_samples // WINDOW, 2 * t_hub)
    return (1, c["inter"], T), (1, T * c["upp"], 1)


def hubert_frames(n: int, cfg) -> int:
    for k, s in zip(cfg["conv_kernels"], cfg["conv_strides"]):
        n = (n - k) // s + 1
    return n


@torch.no_grad()
def pipeline(models: Models, geo: Geometry, audio: np.ndarray, pitch: float = 0, sid: int = 0,
             big_npy=None, index_rate: float = 0.0, volume_envelope: float = 1.0,
             protect: float = 0.33, f0_min=50, f0_max=1100, noises=None, seed: int = 0,
             return_parts=False, f0_method: str = "rmvpe+", inp_f0=None, hop_length: int = 128, crepe_dither=None):
    """VC.pipeline (pipeline.py:289-467) with f0_method="rmvpe+", "fcpe" (models.fcpe_sd) or "mangio-crepe"
    (models.crepe_sd, hop_length, crepe_dither: see oracle/crepe.py), pitch_guidance=1,
    resample_sr=0, f0_file=None.  ``noises`` = list of (z_noise, src_noise) per chunk; drawn from
    torch.manual_seed(seed) in the reference's order (z first, then source) if None."""
    audio = highpass(np.asarray(audio, dtype=np.float64))
    opt_ts = chunk_points(audio, geo)
    audio_pad = np.pad(audio, (geo.t_pad, geo.t_pad), mode="reflect")
    p_len = audio_pad.shape[0] // WINDOW
    if f0_method == "mangio-crepe":                           # pipeline.py:151-152
        from . import crepe as O_crepe
        f0 = O_crepe.get_f0_crepe(models.crepe_sd, audio_pad, f0_min, f0_max, p_len, int(hop_length), crepe_dither)
    elif f0_method == "fcpe":                                 # pipeline.py:169-181
        f0 = O_fcpe.compute_f0(models.fcpe_sd, audio_pad.astype(np.float32), p_len, 0.03)
    else:
        f0 = O_rmvpe.infer_f0(models.rmvpe_sd, models.rmvpe_cfg, audio_pad, 0.03, f0_min, f0_max)
    coarse, f0bak = f0_to_coarse(f0, pitch, f0_min, f0_max, inp_f0, geo.t_pad // SR)
    coarse, f0bak = coarse[:p_len], f0bak[:p_len]
    plan = chunk_plan(audio.shape[0], opt_ts, geo)
    if noises is None:
        gen = torch.Generator().manual_seed(seed)
        noises = []
        for (s, e, fs, fe) in plan:
            zs, ss = noise_shapes(models, e - s)
            noises.append((torch.randn(zs, generator=gen), torch.randn(ss, generator=gen)))
    outs, raw = [], []
    for (s, e, fs, fe), (zn, sn) in zip(plan, noises):
        a1 = vc_chunk(models, audio_pad[s:e], coarse[fs:fe], f0bak[fs:fe].astype(np.float32), sid,
                      big_npy, index_rate, protect, zn, sn)
        raw.append(a1)
        outs.append(a1[geo.t_pad_tgt:-geo.t_pad_tgt])
    audio_opt = np.concatenate(outs)
    if volume_envelope != 1:
        audio_opt = change_rms(audio, SR, audio_opt, geo.tgt_sr, volume_envelope)
    pcm = to_int16(audio_opt)
    if return_parts:
        return pcm, dict(audio_f32=audio_opt, f0=f0bak, coarse=coarse, plan=plan, raw=raw,
                         noises=noises, filtered=audio)
    return pcm
