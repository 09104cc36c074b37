"""Mirror of rvc/infer/pipeline.py: class ``VC`` with the reference's constructor and method
signatures, backed by librvcx.so (HIP, gfx950).  Everything between the float64 input array and
the int16 output array runs on the GPU through the C ABI; there is no CPU fallback.

Differences that are deliberate and documented:
  * ``f0_method``: only "rmvpe+" (and its BASELINE alias "rmvpe") is implemented; anything else
    raises ValueError (the reference hits an accidental UnboundLocalError, pipeline.py:152-183).
  * ``filter_radius`` / ``hop_length`` are accepted and unused, as in the reference for rmvpe+.
  * ``model`` / ``net_g`` are opaque handles (``HubertHandle`` / ``SynthHandle`` from .infer).
"""
from __future__ import annotations

import os

import numpy as np

from .. import _lib

_INDEX_RESIDENT = {}   # id(context) -> (realpath, mtime_ns, size) of the FAISS index whose vectors are in HBM

F0_METHODS = ("rmvpe+", "rmvpe")


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
        self.seed = 0

    # ------------------------------------------------------------------------------------
    def _params(self, pitch, index_rate, volume_envelope, protect, f0_min, f0_max, sid=0):
        p = _lib.Params()
        p.pitch, p.f0_min, p.f0_max = float(pitch), float(f0_min), float(f0_max)
        p.index_rate, p.protect, p.volume_envelope = float(index_rate), float(protect), float(volume_envelope)
        p.sid = int(sid)
        p.x_pad, p.x_query, p.x_center, p.x_max = self.x_pad, self.x_query, self.x_center, self.x_max
        p.seed = int(self.seed)
        return p

    @staticmethod
    def _check_method(f0_method):
        if f0_method not in F0_METHODS:
            raise ValueError(f"f0_method={f0_method!r} is not implemented by rvcx (supported: {F0_METHODS})")

    def get_f0(self, input_audio_path, x, p_len, pitch, f0_method, filter_radius, hop_length, inp_f0=None,
               f0_min=50, f0_max=1100, ctx=None):
        """pipeline.py:132-201.  ``x`` is the reflect-padded, high-passed signal in the reference; here
        the un-padded 16 kHz clip is passed and the library pads/filters it (rvcx_get_f0)."""
        self._check_method(f0_method)
        if inp_f0 is not None:
            raise ValueError("f0 files are not supported (the reference's rvc_infer always passes None)")
        coarse, f0 = ctx.get_f0(np.asarray(x, np.float32), self._params(pitch, 0, 1, 0.5, f0_min, f0_max))
        return coarse[:p_len], f0[:p_len]

    def _load_index(self, ctx, file_index, index_rate):
        """pipeline.py:315-328: index + big_npy, failures swallowed (print, continue without index)."""
        # residency is per context (a new VC is built per request): keyed by (realpath, mtime, size)
        if not (file_index is not None and file_index != "" and os.path.exists(file_index) and index_rate != 0):
            if _INDEX_RESIDENT.get(id(ctx)) is not None:
                ctx.load_index(None)
                _INDEX_RESIDENT[id(ctx)] = None
            return
        try:
            st = os.stat(file_index)
            key = (os.path.realpath(file_index), st.st_mtime_ns, st.st_size)
            if _INDEX_RESIDENT.get(id(ctx)) == key:
                return
            from ..index_io import read_index_vectors
            big_npy = read_index_vectors(file_index)
            ctx.load_index(big_npy)
            _INDEX_RESIDENT[id(ctx)] = key
        except Exception as e:  # noqa: BLE001 -- same degrade-to-None behaviour as the reference
            print(f"Error reading the FAISS index: {e}")
            ctx.load_index(None)
            _INDEX_RESIDENT[id(ctx)] = None

    def pipeline(self, model, net_g, sid, audio, input_audio_path, pitch, f0_method, file_index, index_rate,
                 pitch_guidance, filter_radius, tgt_sr, resample_sr, volume_envelope, version, protect,
                 hop_length, f0_file, f0_min=50, f0_max=1100, noise=None, return_f32=False):
        """pipeline.py:289-467 -> np.ndarray[int16]."""
        self._check_method(f0_method)
        if not pitch_guidance:
            raise ValueError("non-f0 models cannot run in the reference either (generators.py:57-77)")
        if version != "v2":
            raise ValueError("only RVC v2 voice models are supported")
        if f0_file is not None:
            raise ValueError("f0 files are not supported")
        if resample_sr >= self.sample_rate and tgt_sr != resample_sr:
            raise ValueError("resample_sr is hard-wired to 0 by rvc_infer (infer.py:144)")
        ctx = net_g.ctx
        if model.ctx is not ctx:
            raise ValueError("hubert and voice model live on different rvcx contexts")
        self._load_index(ctx, file_index, index_rate)
        p = self._params(pitch, index_rate if _INDEX_RESIDENT.get(id(ctx)) else 0.0, volume_envelope, protect, f0_min, f0_max,
                         sid)
        audio = np.asarray(audio, dtype=np.float32)
        res = ctx.convert_batch(net_g.model_id, [audio], p, None if noise is None else [noise],
                                want_f32=return_f32)
        if return_f32:
            return res[0][0], res[1][0]
        return res[0]
