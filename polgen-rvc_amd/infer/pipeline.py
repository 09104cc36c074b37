"""Mirror of rvc/infer/pipeline.py: class ``VC`` with the reference's constructor and method
signatures, backed by librvcx.so (HIP, gfx950).  Everything between the float64 input array and
the int16 output array runs on the GPU through the C ABI; there is no CPU fallback.

Differences that are deliberate and documented:
  * ``f0_file``: read like the reference (an object with ``.name``; rows "time,f0"), applied on the device.
  * ``f0_method``: "rmvpe+" (with its BASELINE alias "rmvpe"), "fcpe" and "mangio-crepe" are implemented; anything
    else raises ValueError (the reference hits an accidental UnboundLocalError, pipeline.py:152-183).
    "mangio-crepe" is torchcrepe's network and Viterbi decoder restated (csrc/crepe.hip; torchcrepe is not vendored
    with the reference: parity unpinned).  Its weights are torchcrepe's ``assets/full.pth`` state dict, looked up at
    ``CREPE_DIR`` or handed to ``infer.load_crepe``; the +-20 cent dither torchcrepe draws from scipy's global RNG is
    drawn the same way here (``_crepe_dither``) and crosses the ABI as an array.
  * ``filter_radius`` is accepted and unused, as in the reference; ``hop_length`` is used by "mangio-crepe" only.
  * ``model`` / ``net_g`` are opaque handles (``HubertHandle`` / ``SynthHandle`` from .infer); ``index`` is
    the handle ``_load_index`` returns (the vectors live in HBM), ``big_npy`` the matrix or ``True``.
  * keyword-only extras after the reference's parameters (``noise`` / ``z_noise`` / ``src_noise`` replace the
    two ``randn_like`` draws for parity runs; ``return_f32``) never change positional calls.
"""
from __future__ import annotations

import functools
import os
import zlib

import numpy as np

from .. import _lib
from . import _state
from ._state import _INDEX_RESIDENT

# rvc/infer/pipeline.py:14-16 -- resolved against the working directory at import, like the reference
RMVPE_DIR = os.path.join(os.getcwd(), "rvc", "models", "predictors", "rmvpe.pt")
FCPE_DIR = os.path.join(os.getcwd(), "rvc", "models", "predictors", "fcpe.pt")
# torchcrepe reads <site-packages>/torchcrepe/assets/full.pth; a deployment copies that file here (or calls infer.load_crepe)
CREPE_DIR = os.path.join(os.getcwd(), "rvc", "models", "predictors", "crepe_full.pth")

F0_METHODS = ("rmvpe+", "rmvpe", "fcpe", "mangio-crepe")
_F0_CODE = {"rmvpe+": _lib.F0_RMVPE, "rmvpe": _lib.F0_RMVPE, "fcpe": _lib.F0_FCPE, "mangio-crepe": _lib.F0_CREPE}


def _crepe_dither(n: int) -> np.ndarray:
    """torchcrepe.convert.dither: scipy.stats.triang.rvs(c=0.5, loc=-20, scale=40, size=n) on numpy's GLOBAL generator
    (a caller who seeds numpy gets the reference's draws); numpy's own triangular law when scipy is absent."""
    try:
        import scipy.stats
        return scipy.stats.triang.rvs(c=0.5, loc=-20, scale=40, size=int(n)).astype(np.float32)
    except ImportError:
        return np.random.triangular(-20.0, 0.0, 20.0, size=int(n)).astype(np.float32)


def _np(a, dtype=None):
    """numpy view of numpy / torch / list input (torch tensors may live on any device)."""
    if hasattr(a, "detach"):
        a = a.detach().cpu().numpy()
    a = np.asarray(a)
    return a if dtype is None else a.astype(dtype, copy=False)


def _with_ctx_lock(method):
    """Hold the context's lock for the whole method: a mirror call is a SEQUENCE of C calls (lazy predictor load, make the
    index resident, convert) that must not interleave with another thread's sequence on the same context."""
    @functools.wraps(method)
    def wrapper(self, *a, **k):
        net_g = k.get("net_g", a[1] if len(a) > 1 else None)
        ctx = getattr(net_g, "ctx", None) or self._ctx()
        with ctx.lock:
            return method(self, *a, **k)
    return wrapper


class IndexHandle:
    """Stands for ``faiss.read_index(file_index)``: the stored vectors are resident in the context."""

    def __init__(self, ctx, key, ntotal, dim):
        self.ctx, self.key, self.ntotal, self.d = ctx, key, ntotal, dim


class VC:
    def __init__(self, tgt_sr, config):
        # rvc/infer/pipeline.py:66-84
        self.x_pad = config.x_pad
        self.x_query = config.x_query
        self.x_center = config.x_center
        self.x_max = config.x_max
        self.is_half = config.is_half
        self.sample_rate = 16000
        self.window = 160
        self.t_pad = self.sample_rate * self.x_pad
        self.t_pad_tgt = tgt_sr * self.x_pad
        self.t_pad2 = self.t_pad * 2
        self.t_query = self.sample_rate * self.x_query
        self.t_center = self.sample_rate * self.x_center
        self.t_max = self.sample_rate * self.x_max
        self.time_step = self.window / self.sample_rate * 1000
        self.device = config.device
        self.tgt_sr = tgt_sr
        self.seed = None        # None: fresh Philox seed per call (the reference uses torch's global RNG)

    # ------------------------------------------------------------------------------------
    def _ctx(self):
        return _state.context(self.device)

    def _params(self, pitch, index_rate, volume_envelope, protect, f0_min, f0_max, sid=0, f0_method="rmvpe+",
                hop_length=128):
        p = _lib.Params()
        p.pitch, p.f0_min, p.f0_max = float(pitch), float(f0_min), float(f0_max)
        p.index_rate, p.protect, p.volume_envelope = float(index_rate), float(protect), float(volume_envelope)
        p.sid = int(sid)
        p.x_pad, p.x_query, p.x_center, p.x_max = self.x_pad, self.x_query, self.x_center, self.x_max
        p.seed = _state.next_seed() if self.seed is None else int(self.seed)
        p.f0_method = _F0_CODE.get(f0_method, _lib.F0_RMVPE)
        p.hop_length = int(hop_length) if hop_length else 128
        return p

    @staticmethod
    def _check_version(version, net_g):
        """pipeline.py:228-236 / infer.py:91-97: "v1" = HuBERT output layer 9 + final_proj into a 256-wide emb_phone, "v2" =
        layer 12 into a 768-wide one.  The library reads the case off the voice model's input width; the caller's
        ``version`` must agree with it (the reference would fail inside emb_phone with a shape error)."""
        if version not in ("v1", "v2"):
            raise ValueError(f"unknown voice model version {version!r} (v1 or v2)")
        want = getattr(net_g, "input_dim", None)
        if (version, want) in (("v1", 768), ("v2", 256)):        # the two canonical widths (infer.py:92)
            raise ValueError(f"version {version!r} does not match the voice model (emb_phone takes {want} features: "
                             f"{'v1' if want == 256 else 'v2'})")

    @staticmethod
    def _check_method(f0_method):
        if f0_method not in F0_METHODS:
            raise ValueError(f"f0_method={f0_method!r} is not implemented by rvcx (supported: {F0_METHODS})")

    def _ensure_rmvpe(self, ctx=None):
        """pipeline.py:123-126: the predictor is created on first use from rvc/models/predictors/rmvpe.pt."""
        ctx = ctx or self._ctx()
        if not getattr(ctx, "rmvpe_loaded", False):
            from . import infer
            infer.load_rmvpe(self.device, RMVPE_DIR)
        self.model_rmvpe = True
        return ctx

    def _ensure_fcpe(self, ctx=None):
        """pipeline.py:169-178: FCPEF0Predictor(FCPE_DIR, ...) -- built per call by the reference, resident here."""
        ctx = ctx or self._ctx()
        if not getattr(ctx, "fcpe_loaded", False):
            from . import infer
            infer.load_fcpe(self.device, FCPE_DIR)
        return ctx

    def _ensure_crepe(self, ctx=None):
        """torchcrepe.predict loads its packaged weights on first use (torchcrepe.load.model); here from CREPE_DIR."""
        ctx = ctx or self._ctx()
        if not getattr(ctx, "crepe_loaded", False):
            from . import infer
            infer.load_crepe(self.device, CREPE_DIR)
        return ctx

    def _ensure_f0_model(self, f0_method, ctx=None):
        if f0_method == "mangio-crepe":
            return self._ensure_crepe(ctx)
        return self._ensure_fcpe(ctx) if f0_method == "fcpe" else self._ensure_rmvpe(ctx)

    # ------------------------------------------------------------------------------------
    @_with_ctx_lock
    def get_f0_crepe(self, x, f0_min, f0_max, p_len, hop_length, model="full", *, dither=None):
        """pipeline.py:86-117: quantile normalisation, torchcrepe.predict (Viterbi decoder, batches of 2 * hop_length
        frames), resize to p_len -> f0 in Hz (float64, unshifted).  ``model`` names torchcrepe's capacity; the weights that
        are resident decide it here."""
        ctx = self._ensure_crepe()
        x = _np(x, np.float32)
        p_len = p_len or x.shape[0] // int(hop_length)
        p = self._params(0, 0, 1, 0.5, f0_min, f0_max, f0_method="mangio-crepe", hop_length=hop_length)
        if dither is None:
            dither = _crepe_dither(ctx.crepe_frames(x.shape[0], p.hop_length))
        return ctx.get_f0_crepe_x(x, p_len, p, None, dither)[1].astype(np.float64)

    @_with_ctx_lock
    def get_f0_rmvpe(self, x, f0_min=1, f0_max=40000, *args, **kwargs):
        """pipeline.py:119-130 -> f0 in Hz, 1 + len(x)//160 frames; out-of-range frames are 0 (rmvpe+)."""
        ctx = self._ensure_rmvpe()
        f0 = ctx.rmvpe_f0(_np(x, np.float32), 0.03, f0_min, f0_max)[0]
        return f0.astype(np.float64)

    @_with_ctx_lock
    def get_f0(self, input_audio_path, x, p_len, pitch, f0_method, filter_radius, hop_length, inp_f0=None,
               f0_min=50, f0_max=1100, *, crepe_dither=None):
        """pipeline.py:132-201.  ``x`` is the reflect-padded, high-passed signal, as in the reference; returns
        (f0_coarse int array, f0 float array): 1 + len(x)//160 frames for rmvpe+ (the caller truncates to p_len),
        p_len frames for fcpe (FCPEF0Predictor.compute_f0 resizes to p_len itself, FCPE.py:869-877)."""
        self._check_method(f0_method)
        ctx = self._ensure_f0_model(f0_method)
        x = _np(x, np.float32)
        # the whole of get_f0 -- F0 model, pitch shift, the optional f0-file table (pipeline.py:185-191), coarse
        # quantisation -- runs behind the C ABI (rvcx_get_f0_x_ex); rvc_infer never passes inp_f0 (f0_file=None, infer.py:149)
        p = self._params(pitch, 0, 1, 0.5, f0_min, f0_max, f0_method=f0_method, hop_length=hop_length)
        tab = None if inp_f0 is None else _np(inp_f0, np.float32)
        if f0_method == "mangio-crepe":                   # pipeline.py:151-152; p_len frames, like fcpe
            if crepe_dither is None:
                crepe_dither = _crepe_dither(ctx.crepe_frames(x.shape[0], p.hop_length))
            coarse, f0 = ctx.get_f0_crepe_x(x, p_len, p, tab, crepe_dither)
        else:
            coarse, f0 = ctx.get_f0_x_ex(x, p_len, p, tab)
        return coarse.astype(np.int64), f0.astype(np.float64)

    @_with_ctx_lock
    def vc(self, model, net_g, sid, audio0, pitch, pitchf, index, big_npy, index_rate, version, protect, *,
           z_noise=None, src_noise=None):
        """pipeline.py:203-287: one chunk of audio_pad -> np.float32 waveform (un-trimmed)."""
        self._check_version(version, net_g)
        if pitch is None or pitchf is None:
            raise ValueError("non-f0 models cannot run in the reference either (generators.py:57-77)")
        ctx = net_g.ctx
        if model.ctx is not ctx:
            raise ValueError("hubert and voice model live on different rvcx contexts")
        rate = 0.0
        if index is not None and big_npy is not None and index_rate != 0:
            self._make_resident(ctx, index, big_npy)
            rate = float(index_rate)
        audio0 = _np(audio0)
        if audio0.ndim == 2:
            audio0 = audio0.mean(-1)
        sid = int(_np(sid).ravel()[0]) if not isinstance(sid, int) else sid
        seed = _state.next_seed() if self.seed is None else int(self.seed)
        return ctx.vc(net_g.model_id, audio0.astype(np.float32), _np(pitch).ravel(), _np(pitchf).ravel(), sid, rate,
                      float(protect), z_noise, src_noise, seed)

    # ------------------------------------------------------------------------------------
    @staticmethod
    def _make_resident(ctx, index, big_npy):
        """vc() was handed an index the context may not hold (callers other than pipeline())."""
        if isinstance(index, IndexHandle):
            key = index.key
        elif isinstance(big_npy, np.ndarray):
            # an array is identified by what it holds, never by id(): a new array at a recycled id is a new index
            key = ("array", big_npy.shape, str(big_npy.dtype), zlib.crc32(np.ascontiguousarray(big_npy).view(np.uint8)))
        else:
            key = None
        if key is not None and _INDEX_RESIDENT.get(id(ctx)) == key:
            return
        if isinstance(big_npy, np.ndarray):
            ctx.load_index(big_npy)
            _INDEX_RESIDENT[id(ctx)] = key
        else:
            raise ValueError("index vectors are not resident in this context and no matrix was given")

    def _load_index(self, ctx, file_index, index_rate):
        """pipeline.py:315-328: index + big_npy, failures swallowed (print, continue without index).
        Returns (index handle, big_npy stand-in) or (None, None)."""
        # residency is per context (a new VC is built per request): keyed by (realpath, mtime, size)
        if not (file_index is not None and file_index != "" and os.path.exists(file_index) and index_rate != 0):
            if _INDEX_RESIDENT.get(id(ctx)) is not None:
                ctx.load_index(None)
                _INDEX_RESIDENT[id(ctx)] = None
            return None, None
        from ..index_io import UnsupportedIndex, read_index
        try:
            key = _state.file_key(file_index)
            if _INDEX_RESIDENT.get(id(ctx)) != key:
                ix = read_index(file_index)
                big_npy = ix.vectors
                if ix.is_ivf:       # RVC's "IVF{n},Flat" files: searched like faiss does, nprobe from the file
                    if ix.nprobe != 1:
                        raise UnsupportedIndex(f"IVF index with nprobe = {ix.nprobe}: rvcx searches nprobe = 1 only "
                                               "(what RVC training writes)")
                    ctx.load_index_ivf(big_npy, ix.centroids, ix.assign, ix.nprobe)
                else:
                    ctx.load_index(big_npy)
                _INDEX_RESIDENT[id(ctx)] = key
                ctx._index_shape = big_npy.shape
            n, d = getattr(ctx, "_index_shape", (0, 0))
            return IndexHandle(ctx, key, n, d), True
        except UnsupportedIndex:
            # a file faiss would have read and searched: converting WITHOUT the index would silently change the result
            ctx.load_index(None)
            _INDEX_RESIDENT[id(ctx)] = None
            raise
        except Exception as e:  # noqa: BLE001 -- unreadable file: same degrade-to-None behaviour as the reference
            print(f"Error reading the FAISS index: {e}")
            ctx.load_index(None)
            _INDEX_RESIDENT[id(ctx)] = None
            return None, None

    def pipeline(self, model, net_g, sid, audio, input_audio_path, pitch, f0_method, file_index, index_rate,
                 pitch_guidance, filter_radius, tgt_sr, resample_sr, volume_envelope, version, protect,
                 hop_length, f0_file, f0_min=50, f0_max=1100, *, noise=None, return_f32=False, crepe_dither=None):
        """pipeline.py:289-467 -> np.ndarray[int16]."""
        if noise is None:
            noise = getattr(self, "parity_noise", None)      # test hook for callers that cannot pass `noise`
        return self.pipeline_batch(model, net_g, sid, [audio], pitch, f0_method, file_index, index_rate,
                                   pitch_guidance, tgt_sr, resample_sr, volume_envelope, version, protect, f0_file,
                                   f0_min, f0_max, noise=None if noise is None else [noise],
                                   return_f32=return_f32, _single=True, hop_length=hop_length,
                                   crepe_dither=None if crepe_dither is None else [crepe_dither])

    @_with_ctx_lock
    def pipeline_batch(self, model, net_g, sid, audios, pitch, f0_method, file_index, index_rate, pitch_guidance,
                       tgt_sr, resample_sr, volume_envelope, version, protect, f0_file=None, f0_min=50, f0_max=1100,
                       *, noise=None, return_f32=False, _single=False, hop_length=128, crepe_dither=None):
        """VC.pipeline over a list of utterances in one call (the reference lists batch conversion as not
        done, TODO.md:11): equal-length clips run through the networks together."""
        self._check_method(f0_method)
        if not pitch_guidance:
            raise ValueError("non-f0 models cannot run in the reference either (generators.py:57-77)")
        self._check_version(version, net_g)
        inp_f0 = None
        if f0_file and hasattr(f0_file, "name"):              # pipeline.py:349-360 (failures are printed and ignored)
            try:
                with open(f0_file.name, "r") as f:
                    lines = f.read().strip("\n").split("\n")
                inp_f0 = np.array([[float(i) for i in line.split(",")] for line in lines], dtype="float32")
            except Exception as e:  # noqa: BLE001
                print(f"Error reading the F0 file: {e}")
        ctx = net_g.ctx
        if model.ctx is not ctx:
            raise ValueError("hubert and voice model live on different rvcx contexts")
        self._ensure_f0_model(f0_method, ctx)
        index, _ = self._load_index(ctx, file_index, index_rate)
        p = self._params(pitch, index_rate if index is not None else 0.0, volume_envelope, protect, f0_min, f0_max,
                         int(_np(sid).ravel()[0]) if not isinstance(sid, int) else sid, f0_method=f0_method,
                         hop_length=hop_length)
        # pipeline.py:453-454: librosa.resample(audio_opt, orig_sr=tgt_sr, target_sr=resample_sr) ahead of the peak
        # normalisation -- on the device (csrc/audio.hip); rvc_infer passes 0 (infer.py:144)
        p.resample_sr = int(resample_sr) if (resample_sr >= self.sample_rate and tgt_sr != resample_sr) else 0
        # float64 stays float64 across the ABI (filtfilt then sees what the reference's sees); else float32
        clips = [a if _np(a).dtype == np.float64 else _np(a, np.float32) for a in map(_np, audios)]
        if not all(c.dtype == clips[0].dtype for c in clips):
            clips = [c.astype(np.float64) for c in clips]
        if f0_method == "mangio-crepe" and crepe_dither is None:     # one value per frame of each PADDED clip, drawn in order
            crepe_dither = [_crepe_dither(ctx.crepe_frames(c.shape[0] + 2 * self.t_pad, p.hop_length)) for c in clips]
        res = ctx.convert_batch(net_g.model_id, clips, p, noise, want_f32=return_f32,
                                inp_f0=None if inp_f0 is None else [inp_f0] * len(clips),
                                crepe_dither=crepe_dither if f0_method == "mangio-crepe" else None)
        if _single:
            return (res[0][0], res[1][0]) if return_f32 else res[0]
        return res
