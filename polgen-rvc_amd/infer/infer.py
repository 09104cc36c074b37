"""Mirror of rvc/infer/infer.py: ``Config``, ``load_hubert``, ``get_vc``, ``rvc_infer`` with the
reference's signatures (call sites: rvc/scripts/voice_conversion.py:71-96,
rvc/scripts/edge_tts_conversion.py:79-104), backed by librvcx.so.
"""
from __future__ import annotations

import os
import wave
from multiprocessing import cpu_count

import numpy as np

from .. import _lib, weights
from .pipeline import VC

from . import _state
from ._state import _CTX, _RESIDENT, _SYNTHS   # noqa: F401  (tests reach in)

# ---- resident-asset cache -------------------------------------------------------------
# The reference's scripts reload HuBERT, the voice model and RMVPE on every request
# (rvc/scripts/voice_conversion.py:71-75,98-100).  Here a checkpoint path that was loaded before, and whose
# (realpath, mtime, size) has not changed, maps to the models already folded and resident in HBM.
MAX_RESIDENT_SYNTHS = 8   # voice models kept per process (a 48 k model is ~230 MB of folded weights incl. fp16 images)

_file_key = _state.file_key
_dev_index = _state.dev_index
_context = _state.context


@_state.locked
def clear_cache():
    """Drop every cached handle (voice models are unloaded once their callers released them too)."""
    _RESIDENT.clear()
    _SYNTHS.clear()


class Config:
    """rvc/infer/infer.py:12-63.  On an MI355X under ROCm the reference resolves to device="cuda",
    is_half=False and the (1, 6, 38, 41) chunk geometry; rvcx computes in fp32 and has no CPU path."""

    def __init__(self, device="cuda:0"):
        self.device = device
        self.is_half = False
        self.n_cpu = cpu_count()
        # infer.py:49-63 (_configure_gpu): name and GiB of the device, read from the HIP runtime; None without a GPU
        # (the reference leaves both None on its CPU branch).  is_half stays False: rvcx computes in fp32.
        name, total = _lib.device_info(_dev_index(device))
        self.gpu_name = name
        self.gpu_mem = None if total is None else int(total / 1024 / 1024 / 1024 + 0.4)
        self.x_pad, self.x_query, self.x_center, self.x_max = (1, 6, 38, 41)


class HubertHandle:
    def __init__(self, ctx, cfg):
        self.ctx, self.cfg = ctx, cfg


class SynthHandle:
    def __init__(self, ctx, model_id, cfg, input_dim=768):
        self.ctx, self.model_id, self.cfg, self.input_dim = ctx, model_id, cfg, input_dim

    def __del__(self):
        try:
            _lib.lib().rvcx_unload_synth(self.ctx._h, self.model_id)
        except Exception:
            pass


def _torch_load(path):
    import torch  # PyTorch is used for un-pickling checkpoints only
    return torch.load(path, map_location="cpu", weights_only=True)


@_state.locked_dev
def load_hubert(device, is_half, model_path, state=None, cfg=None):
    """rvc/infer/infer.py:67-74.  ``model_path`` is a fairseq ``hubert_base.pt`` (its ``"model"``
    tensor dict is read with a restricted unpickler); alternatively pass the state dict directly."""
    ctx = _context(device)
    slot, key = (_dev_index(device), "hubert"), None
    if state is None:
        key = _file_key(model_path)
        if slot in _RESIDENT and _RESIDENT[slot][0] == key:
            return _RESIDENT[slot][1]
        from ..ckpt_io import load_fairseq_hubert
        state = load_fairseq_hubert(model_path)
    cfg = cfg or weights.hubert_cfg_from_state(state)
    ctx.load_hubert(weights.hubert_cfg_struct(cfg), state)
    handle = HubertHandle(ctx, cfg)
    _RESIDENT.pop(slot, None)
    if key is not None:
        _RESIDENT[slot] = (key, handle)
    return handle


@_state.locked_dev
def load_rmvpe(device, model_path=None, state=None, cfg=None):
    """RMVPE0Predictor.__init__ (rvc/lib/predictors/RMVPE.py:442-459); the reference loads it lazily
    inside VC.get_f0_rmvpe from rvc/models/predictors/rmvpe.pt (pipeline.py:123-126)."""
    ctx = _context(device)
    slot, key = (_dev_index(device), "rmvpe"), None
    if state is None:
        key = _file_key(model_path)
        if slot in _RESIDENT and _RESIDENT[slot][0] == key and getattr(ctx, "rmvpe_loaded", False):
            return
        state = _torch_load(model_path)
    ctx.load_rmvpe(weights.rmvpe_cfg_struct(cfg or weights.rmvpe_cfg_from_state(state)), state)
    ctx.rmvpe_loaded = True
    _RESIDENT.pop(slot, None)
    if key is not None:
        _RESIDENT[slot] = (key, True)


@_state.locked_dev
def load_crepe(device, model_path=None, state=None):
    """torchcrepe.load.model (called inside torchcrepe.predict, rvc/infer/pipeline.py:96): the state dict of
    torchcrepe's model.Crepe -- its packaged ``assets/full.pth`` (or ``tiny.pth``), copied to ``model_path``, or given
    as ``state``.  The capacity is read off the tensor shapes.  Stays resident like the other predictors."""
    ctx = _context(device)
    slot, key = (_dev_index(device), "crepe"), None
    if state is None:
        if not model_path or not os.path.exists(model_path):
            raise FileNotFoundError(f"mangio-crepe needs torchcrepe's weights (assets/full.pth) at {model_path!r} "
                                    "or through infer.load_crepe(device, state=...)")
        key = _file_key(model_path)
        if slot in _RESIDENT and _RESIDENT[slot][0] == key and getattr(ctx, "crepe_loaded", False):
            return
        state = _torch_load(model_path)
    ctx.load_crepe(state)
    ctx.crepe_loaded = True
    _RESIDENT.pop(slot, None)
    if key is not None:
        _RESIDENT[slot] = (key, True)


@_state.locked_dev
def load_fcpe(device, model_path=None, cpt=None):
    """FCPEF0Predictor.__init__ -> FCPEInfer.__init__ (rvc/lib/predictors/FCPE.py:806-826, 708-736): fcpe.pt is
    ``{"config": {...}, "model": state_dict}``; the reference builds it inside VC.get_f0 on every call from
    rvc/models/predictors/fcpe.pt (pipeline.py:169-181) and drops it again -- here it stays resident."""
    ctx = _context(device)
    slot, key = (_dev_index(device), "fcpe"), None
    if cpt is None:
        key = _file_key(model_path)
        if slot in _RESIDENT and _RESIDENT[slot][0] == key and getattr(ctx, "fcpe_loaded", False):
            return
        cpt = _torch_load(model_path)
    if "model" not in cpt:
        raise ValueError("Invalid fcpe checkpoint: no 'model' state dict (FCPE.py:734)")
    model_cfg = dict(cpt.get("config", {}).get("model", {}))
    if model_cfg.get("use_siren") or model_cfg.get("use_full"):
        raise ValueError("Siren / full FCPE models are not supported (nor by the reference, FCPE.py:573-576)")
    if model_cfg.get("confidence"):
        raise ValueError("fcpe checkpoints with confidence=True make the reference return a tuple it cannot index")
    ctx.load_fcpe(weights.fcpe_cfg_struct(weights.fcpe_cfg_from_state(cpt["model"], cpt.get("config"))), cpt["model"])
    ctx.fcpe_loaded = True
    _RESIDENT.pop(slot, None)
    if key is not None:
        _RESIDENT[slot] = (key, True)


def get_vc(device, is_half, config, model_path, cpt=None):
    """rvc/infer/infer.py:78-105 -> (cpt, version, net_g, tgt_sr, vc).  A ``model_path`` seen before (same
    realpath, mtime and size) returns the voice model already resident in HBM; the returned ``cpt`` then
    carries the checkpoint's metadata (config, version, f0, ...) without the weight tensors.

    Locking: the process-wide table lock is held for the ``_SYNTHS`` lookup and for the insert only -- un-pickling and
    validation (seconds for a real checkpoint) run outside it, so requests of other threads (whose ``context()`` /
    table reads used to queue behind a model load) go on; two threads that load the same new file both load it and the
    second finds the first's entry at the re-check (its own copy is unloaded with its handle).  ``ctx.load_synth``
    serialises against conversions on the context's C mutex, as every entry point does."""
    skey = None

    def hit():
        light, net_g = _SYNTHS.pop(skey)
        _SYNTHS[skey] = (light, net_g)              # most recently used
        return dict(light), light.get("version", "v1"), net_g, light["config"][-1], VC(light["config"][-1], config)
    if cpt is None:
        skey = (_dev_index(device), _file_key(model_path))
        with _state.LOCK:
            if skey in _SYNTHS:
                return hit()
        cpt = _torch_load(model_path)
    if "config" not in cpt or "weight" not in cpt:
        raise ValueError(f"Invalid format for {model_path}. Use a voice model trained with RVC v2.")     # the reference's text (infer.py:84)
    tgt_sr = cpt["config"][-1]
    cpt["config"][-3] = cpt["weight"]["emb_g.weight"].shape[0]
    pitch_guidance = cpt.get("f0", 1)
    version = cpt.get("version", "v1")
    if not pitch_guidance:
        raise ValueError("rvcx supports voice models with pitch guidance (f0=1); the reference cannot run the others either "
                         "(generators.py:57-77)")
    if version not in ("v1", "v2"):
        raise ValueError(f"unknown voice model version {version!r}")
    # infer.py:91-97: input_dim = 768 if version == "v2" else 256 -- read off the tensor, checked against the version
    input_dim = int(cpt["weight"]["enc_p.emb_phone.weight"].shape[1])
    if (version, input_dim) in (("v1", 768), ("v2", 256)):
        raise ValueError(f"checkpoint says version {version!r} but enc_p.emb_phone takes {input_dim} features")
    state = weights.strip_enc_q(cpt["weight"])
    cfg_struct = weights.synth_cfg_struct(cpt["config"], input_dim)
    ctx = _context(device)
    mid = ctx.load_synth(cfg_struct, state)         # outside the table lock (the context's own mutex orders it)
    net_g = SynthHandle(ctx, mid, list(cpt["config"]), input_dim)
    if skey is not None:
        with _state.LOCK:
            if skey in _SYNTHS:                     # another thread loaded the same file meanwhile: keep theirs,
                del net_g                           # ours is unloaded with its handle
                return hit()
            _SYNTHS[skey] = ({k: v for k, v in cpt.items() if k != "weight"}, net_g)
            while len(_SYNTHS) > MAX_RESIDENT_SYNTHS:
                _SYNTHS.pop(next(iter(_SYNTHS)))    # least recently used; unloads when the caller let go too
    return cpt, version, net_g, tgt_sr, VC(tgt_sr, config)


def load_audio(file, sample_rate, *, device=None):
    """rvc/lib/my_utils.py:5-16 (`device`: keyword-only extra, default = the GPU this process already serves): read -> mono mean -> resample -> flatten (float64).  Decoding: infer/audio.py
    (soundfile when installed, RIFF/WAVE otherwise).  The mono mix and the rate conversion run on the GPU in one pass
    (``rvcx_resample_f64``): a Kaiser-windowed sinc designed to soxr_hq's published targets (librosa's default res_type;
    libsoxr's coefficients are unpublished, so this edge is "parity unpinned, within the published spec": oracle/audio.py)"""
    try:
        file = file.strip(" ").strip('"').strip("\n").strip('"').strip(" ")
        from .audio import read_audio
        audio, sr = read_audio(file)
        if sr != sample_rate:
            ctx = _context(device)
            with ctx.lock:
                audio = ctx.resample(audio, sr, sample_rate)
        elif len(audio.shape) > 1:
            audio = np.asarray(audio, np.float64).mean(axis=1)        # librosa.to_mono(audio.T)
    except Exception as error:
        raise RuntimeError(f"An error occurred loading the audio: {error}")
    return np.asarray(audio, np.float64).flatten()


def rvc_infer(index_path, index_rate, input_path, output_path, pitch, f0_method, cpt, version, net_g,
              filter_radius, tgt_sr, volume_envelope, protect, hop_length, vc, hubert_model, f0_min=50,
              f0_max=1100):
    """rvc/infer/infer.py:109-153: load -> vc.pipeline -> WAV bytes whatever the extension -- except ".flac", which gets a
    real FLAC stream (audio.write_output)."""
    from .audio import write_output
    audio = load_audio(input_path, 16000)
    pitch_guidance = cpt.get("f0", 1)
    audio_opt = vc.pipeline(hubert_model, net_g, 0, audio, input_path, pitch, f0_method, index_path, index_rate,
                            pitch_guidance, filter_radius, tgt_sr, 0, volume_envelope, version, protect,
                            hop_length, f0_file=None, f0_min=f0_min, f0_max=f0_max)
    # infer.py:153: sf.write(output_path, audio_opt, tgt_sr, format="WAV") -- a WAV whatever the extension says
    write_output(output_path, audio_opt, tgt_sr)
