"""Mirror of rvc/infer/infer.py: ``Config``, ``load_hubert``, ``get_vc``, ``rvc_infer`` with the
reference's signatures (call sites: rvc/scripts/voice_conversion.py:71-96,
rvc/scripts/edge_tts_conversion.py:79-104), backed by librvcx.so.
"""
from __future__ import annotations

import wave
from multiprocessing import cpu_count

import numpy as np

from .. import _lib, weights
from .pipeline import VC

_CTX = {}


def _context(device) -> "_lib.Context":
    """One rvcx context per GPU, created on first use."""
    idx = 0
    if isinstance(device, str) and ":" in device:
        idx = int(device.split(":")[1])
    elif isinstance(device, int):
        idx = device
    if idx not in _CTX:
        _CTX[idx] = _lib.Context(idx)
    return _CTX[idx]


class Config:
    """rvc/infer/infer.py:12-63.  On an MI355X under ROCm the reference resolves to device="cuda",
    is_half=False and the (1, 6, 38, 41) chunk geometry; rvcx computes in fp32 and has no CPU path."""

    def __init__(self, device="cuda:0"):
        self.device = device
        self.is_half = False
        self.n_cpu = cpu_count()
        self.gpu_name = "AMD Instinct MI355X"
        self.gpu_mem = 288
        self.x_pad, self.x_query, self.x_center, self.x_max = (1, 6, 38, 41)


class HubertHandle:
    def __init__(self, ctx, cfg):
        self.ctx, self.cfg = ctx, cfg


class SynthHandle:
    def __init__(self, ctx, model_id, cfg):
        self.ctx, self.model_id, self.cfg = ctx, model_id, cfg

    def __del__(self):
        try:
            _lib.lib().rvcx_unload_synth(self.ctx._h, self.model_id)
        except Exception:
            pass


def _torch_load(path):
    import torch  # PyTorch is used for un-pickling checkpoints only
    return torch.load(path, map_location="cpu", weights_only=True)


def load_hubert(device, is_half, model_path, state=None, cfg=None):
    """rvc/infer/infer.py:67-74.  ``model_path`` is a fairseq ``hubert_base.pt`` (its ``"model"``
    tensor dict is read with a restricted unpickler); alternatively pass the state dict directly."""
    from .. import synthetic
    ctx = _context(device)
    if state is None:
        from ..ckpt_io import load_fairseq_hubert
        state = load_fairseq_hubert(model_path)
    cfg = cfg or synthetic.HUBERT_CFG_BASE
    ctx.load_hubert(weights.hubert_cfg_struct(cfg), state)
    return HubertHandle(ctx, cfg)


def load_rmvpe(device, model_path=None, state=None, cfg=None):
    """RMVPE0Predictor.__init__ (rvc/lib/predictors/RMVPE.py:442-459); the reference loads it lazily
    inside VC.get_f0_rmvpe from rvc/models/predictors/rmvpe.pt (pipeline.py:123-126)."""
    from .. import synthetic
    ctx = _context(device)
    if state is None:
        state = _torch_load(model_path)
    ctx.load_rmvpe(weights.rmvpe_cfg_struct(cfg or synthetic.RMVPE_CFG_FULL), state)


def get_vc(device, is_half, config, model_path, cpt=None):
    """rvc/infer/infer.py:78-105 -> (cpt, version, net_g, tgt_sr, vc)."""
    if cpt is None:
        cpt = _torch_load(model_path)
    if "config" not in cpt or "weight" not in cpt:
        raise ValueError(f"Invalid format for {model_path}. Use a voice model trained with RVC v2.")
    tgt_sr = cpt["config"][-1]
    cpt["config"][-3] = cpt["weight"]["emb_g.weight"].shape[0]
    pitch_guidance = cpt.get("f0", 1)
    version = cpt.get("version", "v1")
    if version != "v2" or not pitch_guidance:
        raise ValueError("rvcx supports RVC v2 voice models with pitch guidance (f0=1)")
    input_dim = int(cpt["weight"]["enc_p.emb_phone.weight"].shape[1])
    ctx = _context(device)
    state = weights.strip_enc_q(cpt["weight"])
    mid = ctx.load_synth(weights.synth_cfg_struct(cpt["config"], input_dim), state)
    net_g = SynthHandle(ctx, mid, list(cpt["config"]))
    vc = VC(tgt_sr, config)
    return cpt, version, net_g, tgt_sr, vc


def load_audio(file, sample_rate):
    """rvc/lib/my_utils.py:5-16 for PCM WAV input at the target rate (soundfile/librosa absent here)."""
    try:
        file = file.strip(" ").strip('"').strip("\n").strip('"').strip(" ")
        from scipy.io import wavfile
        sr, audio = wavfile.read(file)
        if audio.dtype == np.int16:
            audio = audio.astype(np.float64) / 32768.0
        elif audio.dtype == np.int32:
            audio = audio.astype(np.float64) / 2147483648.0
        else:
            audio = audio.astype(np.float64)
        if audio.ndim > 1:
            audio = audio.mean(axis=1)
        if sr != sample_rate:
            raise ValueError(f"input is {sr} Hz; resampling is outside the accelerated path (SURVEY §8f)")
    except Exception as error:
        raise RuntimeError(f"An error occurred loading the audio: {error}")
    return audio.flatten()


def rvc_infer(index_path, index_rate, input_path, output_path, pitch, f0_method, cpt, version, net_g,
              filter_radius, tgt_sr, volume_envelope, protect, hop_length, vc, hubert_model, f0_min=50,
              f0_max=1100):
    """rvc/infer/infer.py:109-153: load -> vc.pipeline -> WAV (whatever the extension)."""
    from scipy.io import wavfile
    audio = load_audio(input_path, 16000)
    pitch_guidance = cpt.get("f0", 1)
    audio_opt = vc.pipeline(hubert_model, net_g, 0, audio, input_path, pitch, f0_method, index_path, index_rate,
                            pitch_guidance, filter_radius, tgt_sr, 0, volume_envelope, version, protect,
                            hop_length, f0_file=None, f0_min=f0_min, f0_max=f0_max)
    wavfile.write(output_path, tgt_sr, audio_opt)
