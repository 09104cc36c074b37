"""ctypes binding of librvcx.so (C ABI declared in include/rvcx.h).

Loading is strict: a missing library is an ImportError-grade failure (``RvcxError``) -- there
is no CPU fallback anywhere in the product path.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RVCX_LIBRARY") or os.path.join(_HERE, "librvcx.so")   # RVCX_LIBRARY: A/B runs of two builds


class RvcxError(RuntimeError):
    pass


class Tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int32),
                ("ndim", C.c_int32), ("shape", C.c_int64 * 4)]


class SynthCfg(C.Structure):
    _fields_ = [("inter_channels", C.c_int32), ("hidden_channels", C.c_int32),
                ("filter_channels", C.c_int32), ("n_heads", C.c_int32), ("n_layers", C.c_int32),
                ("kernel_size", C.c_int32), ("n_resblocks", C.c_int32),
                ("res_kernels", C.c_int32 * 4), ("res_dilations", (C.c_int32 * 3) * 4),
                ("n_ups", C.c_int32), ("up_rates", C.c_int32 * 6), ("up_kernels", C.c_int32 * 6),
                ("up_initial_channel", C.c_int32), ("spk_embed_dim", C.c_int32),
                ("gin_channels", C.c_int32), ("sr", C.c_int32), ("input_dim", C.c_int32)]


class RmvpeCfg(C.Structure):
    _fields_ = [("n_blocks", C.c_int32), ("en_de_layers", C.c_int32), ("inter_layers", C.c_int32),
                ("en_out_channels", C.c_int32)]


class FcpeCfg(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("n_chans", C.c_int32), ("input_channel", C.c_int32),
                ("out_dims", C.c_int32), ("heads", C.c_int32), ("dim_head", C.c_int32),
                ("nb_features", C.c_int32), ("dw_kernel", C.c_int32), ("mel_fmin", C.c_float),
                ("mel_fmax", C.c_float)]


class HubertCfg(C.Structure):
    _fields_ = [("conv_dim", C.c_int32), ("n_conv", C.c_int32), ("conv_kernels", C.c_int32 * 8),
                ("conv_strides", C.c_int32 * 8), ("embed_dim", C.c_int32), ("ffn_dim", C.c_int32),
                ("heads", C.c_int32), ("layers", C.c_int32), ("pos_kernel", C.c_int32),
                ("pos_groups", C.c_int32)]


class Params(C.Structure):
    _fields_ = [("pitch", C.c_float), ("f0_min", C.c_float), ("f0_max", C.c_float),
                ("index_rate", C.c_float), ("protect", C.c_float), ("volume_envelope", C.c_float),
                ("sid", C.c_int32), ("x_pad", C.c_int32), ("x_query", C.c_int32),
                ("x_center", C.c_int32), ("x_max", C.c_int32), ("seed", C.c_uint64),
                ("f0_method", C.c_int32), ("resample_sr", C.c_int32), ("hop_length", C.c_int32), ("reserved", C.c_int32)]


class UttExtra(C.Structure):
    _fields_ = [("inp_f0", C.POINTER(C.c_float)), ("inp_f0_rows", C.c_int32), ("reserved", C.c_int32),
                ("crepe_dither", C.POINTER(C.c_float)), ("crepe_dither_n", C.c_int64)]


F0_RMVPE, F0_FCPE, F0_CREPE = 0, 1, 2        # rvcx_params.f0_method


_lib = None

# every symbol include/rvcx.h declares (tests/test_abi.py checks the .so exports all of them)
SYMBOLS = [
    "rvcx_create", "rvcx_destroy", "rvcx_last_error", "rvcx_version", "rvcx_load_hubert",
    "rvcx_load_rmvpe", "rvcx_index_exhaustive", "rvcx_load_crepe", "rvcx_crepe_frames", "rvcx_crepe_predict", "rvcx_op_crepe_decode", "rvcx_get_f0_crepe_x", "rvcx_load_fcpe", "rvcx_fcpe_f0", "rvcx_fcpe_frames", "rvcx_get_f0_fcpe_x", "rvcx_op_fcpe_post", "rvcx_load_synth", "rvcx_unload_synth", "rvcx_load_index", "rvcx_load_index_ivf",
    "rvcx_weights_regions", "rvcx_weights_adopt", "rvcx_weights_clone", "rvcx_rmvpe_f0", "rvcx_rmvpe_frames", "rvcx_rmvpe_mel", "rvcx_synth_infer_taps", "rvcx_hubert_features",
    "rvcx_hubert_frames", "rvcx_synth_infer", "rvcx_synth_infer_window", "rvcx_synth_dec_rf", "rvcx_synth_upp", "rvcx_index_blend",
    "rvcx_out_len", "rvcx_convert_batch", "rvcx_convert_batch_f64", "rvcx_micro_batch", "rvcx_bucket_length", "rvcx_last_micro_batches", "rvcx_noise_len",
    "rvcx_get_f0", "rvcx_get_f0_x", "rvcx_vc", "rvcx_vc_frames", "rvcx_last_timing",
    "rvcx_flop_counter", "rvcx_fp32_reruns", "rvcx_mem_info", "rvcx_conv_profile", "rvcx_conv_profile_csv", "rvcx_stream", "rvcx_op_conv1d", "rvcx_op_resblock_pair", "rvcx_bench_resblock_pair", "rvcx_bench_conv1d", "rvcx_conv_override", "rvcx_op_convtranspose1d",
    "rvcx_op_conv2d3x3", "rvcx_op_convblock2d", "rvcx_op_convtranspose2d", "rvcx_op_attention", "rvcx_op_layernorm_c",
    "rvcx_op_bigru", "rvcx_op_highpass", "rvcx_convert_batch_ex", "rvcx_get_f0_x_ex", "rvcx_fp32_layers", "rvcx_fp32_pinned",
    "rvcx_gru_fallbacks", "rvcx_gru_publish_probe", "rvcx_debug_inject", "rvcx_f0_file_track", "rvcx_op_gemm_tm", "rvcx_op_layernorm_tm",
    "rvcx_resample_len", "rvcx_resample_f64", "rvcx_resample_f64_kind", "rvcx_bench_gemm", "rvcx_device_info",
    "rvcx_op_resblock3", "rvcx_flac_encode_bound", "rvcx_flac_encode_s16", "rvcx_flac_info", "rvcx_flac_decode_s32", "rvcx_flac_last_error",
]


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RvcxError(f"{LIB_PATH} not built (run `make` / __graft_entry__.build()); "
                            "rvcx has no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.rvcx_last_error.restype = C.c_char_p
        _lib.rvcx_version.restype = C.c_char_p
        _lib.rvcx_flop_counter.restype = C.c_double
        _lib.rvcx_stream.restype = C.c_void_p
        _lib.rvcx_conv_profile_csv.restype = C.c_char_p
        _lib.rvcx_out_len.restype = C.c_int64
        _lib.rvcx_fp32_reruns.restype = C.c_int64
        _lib.rvcx_fp32_layers.restype = C.c_int64
        _lib.rvcx_gru_fallbacks.restype = C.c_int64
        _lib.rvcx_bucket_length.restype = C.c_int64
        _lib.rvcx_resample_len.restype = C.c_int64
        _lib.rvcx_noise_len.restype = C.c_int64
        _lib.rvcx_crepe_frames.restype = C.c_int64
        _lib.rvcx_index_exhaustive.restype = C.c_int64
        _lib.rvcx_flac_encode_bound.restype = C.c_int64
        _lib.rvcx_flac_encode_s16.restype = C.c_int64
        _lib.rvcx_flac_decode_s32.restype = C.c_int64
        _lib.rvcx_flac_last_error.restype = C.c_char_p
        _lib.rvcx_flac_encode_bound.argtypes = [C.c_int64, C.c_int]
        _lib.rvcx_flac_encode_s16.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int64]
        _lib.rvcx_flac_info.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                        C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        _lib.rvcx_flac_decode_s32.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]
    return _lib


_device_info_cache = {}


def device_info(device: int = 0):
    """(name, total bytes) of GPU `device`, or (None, None) when no HIP device is visible (asked once per device: Config()
    is built per request by the reference's callers)"""
    if device in _device_info_cache:
        return _device_info_cache[device]
    _device_info_cache[device] = _device_info(device)
    return _device_info_cache[device]


def _device_info(device: int = 0):
    name = C.create_string_buffer(256)
    total = C.c_int64(0)
    if lib().rvcx_device_info(int(device), name, 256, C.byref(total)) != 0:
        return None, None
    return name.value.decode(), int(total.value)


def flac_encode(pcm, sample_rate: int) -> bytes:
    """int16 PCM, (frames,) or (frames, channels) -> the bytes of a FLAC stream (host code, csrc/flac.hip)"""
    a = np.ascontiguousarray(pcm)
    if a.dtype != np.int16 or a.ndim not in (1, 2):
        raise RvcxError("flac_encode: int16 array of shape (frames,) or (frames, channels) expected")
    frames, ch = a.shape[0], (1 if a.ndim == 1 else a.shape[1])
    cap = lib().rvcx_flac_encode_bound(frames, ch)
    out = np.empty(max(int(cap), 64), np.uint8)
    n = lib().rvcx_flac_encode_s16(a.ctypes.data, frames, ch, int(sample_rate), out.ctypes.data, out.shape[0])
    if n < 0:
        raise RvcxError((lib().rvcx_flac_last_error() or b"").decode())
    return out[:n].tobytes()


def flac_decode(data: bytes):
    """bytes of a FLAC stream -> (int32 array (frames,) or (frames, channels), sample rate, bits per sample)"""
    buf = np.frombuffer(data, np.uint8)
    frames, ch, sr, bits = C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
    if lib().rvcx_flac_info(buf.ctypes.data, buf.shape[0], C.byref(frames), C.byref(ch), C.byref(sr), C.byref(bits)) != 0:
        raise RvcxError((lib().rvcx_flac_last_error() or b"").decode())
    # rvcx_flac_info counts the frames itself when STREAMINFO leaves the total at 0 (streamed encodes) and refuses totals
    # that are implausible for the byte count, so this allocation is exact and bounded by the stream's size
    total = max(1, int(frames.value))
    out = np.empty((total, ch.value), np.int32)
    n = lib().rvcx_flac_decode_s32(buf.ctypes.data, buf.shape[0], out.ctypes.data, out.size)
    if n < 0:
        raise RvcxError((lib().rvcx_flac_last_error() or b"").decode())
    out = out[:n]
    return (out[:, 0].copy() if ch.value == 1 else out.copy()), int(sr.value), int(bits.value)


def f0_file_track(inp_f0) -> np.ndarray:
    """host-side piece of VC.get_f0's f0-file branch as the library computes it (no GPU needed)"""
    tab = np.ascontiguousarray(inp_f0, dtype=np.float32).reshape(-1, 2)
    out = np.empty(65536, np.float64)
    n = lib().rvcx_f0_file_track(tab.ctypes.data_as(C.POINTER(C.c_float)), tab.shape[0],
                                 out.ctypes.data_as(C.POINTER(C.c_double)), out.shape[0])
    if n < 0:
        raise RvcxError("f0_file_track: " + (lib().rvcx_last_error(None) or b"").decode())
    return out[:n].copy()


def _p(a: Optional[np.ndarray], ctype=C.c_float):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ctype))


def f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a) -> Optional[np.ndarray]:
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def make_table(state: dict):
    """dict name -> numpy/torch tensor  ->  (ctypes Tensor array, keep-alive list)."""
    keep, items = [], []
    for name, t in state.items():
        if hasattr(t, "detach"):
            t = t.detach().cpu().numpy()
        a = np.asarray(t)
        if a.dtype == np.float32:
            dt = 0
        elif a.dtype == np.float16:
            dt = 1
        elif a.dtype == np.int64:
            dt = 2
        else:
            a = a.astype(np.float32)
            dt = 0
        a = np.ascontiguousarray(a)
        if a.ndim > 4:
            raise RvcxError(f"tensor {name} has {a.ndim} dims")
        keep.append(a)
        nm = name.encode()
        keep.append(nm)
        shp = (C.c_int64 * 4)(*(list(a.shape) + [1] * (4 - a.ndim)))
        items.append(Tensor(nm, a.ctypes.data, dt, a.ndim, shp))
    arr = (Tensor * len(items))(*items)
    return arr, keep


class Context:
    """One rvcx context (= one GPU)."""

    regions_on_device = True       # weights_regions() returns device pointers (dist.broadcast_weights checks this)

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        rc = lib().rvcx_create(int(device), C.byref(self._h))
        if rc != 0:
            raise RvcxError("rvcx_create: " + (lib().rvcx_last_error(None) or b"").decode())
        self.device = device
        # Every C entry point takes the context's own mutex (csrc/api.hip), so threads that share this object queue
        # instead of racing.  Multi-call sequences that must not interleave -- "make this index resident, then convert
        # with it" in the mirror's VC.pipeline -- hold this lock around the whole sequence.
        self.lock = threading.RLock()

    def close(self):
        if getattr(self, "_h", None):
            lib().rvcx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise RvcxError(f"{what}: " + (lib().rvcx_last_error(self._h) or b"").decode())

    # ---- kernel-level ops -------------------------------------------------------------
    def conv1d(self, x, w, bias=None, res=None, stride=1, dil=1, pad_left=0, Tout=None, groups=1,
               pre_lrelu=None, act=0, act_slope=0.0, lens_in=None, lens_out=None):
        x, w = f32(x), f32(w)
        B, Cin, Tin = x.shape
        Cout, _, K = w.shape
        if Tout is None:
            Tout = (Tin + 2 * pad_left - dil * (K - 1) - 1) // stride + 1
        y = np.empty((B, Cout, Tout), np.float32)
        bias = None if bias is None else f32(bias)
        res = None if res is None else f32(res)
        li, lo = i32(lens_in), i32(lens_out)
        self._ck(lib().rvcx_op_conv1d(self._h, _p(x), _p(w), _p(bias), _p(res), _p(y), B, Cin, Tin, Cout, K,
                                      stride, dil, pad_left, Tout, groups,
                                      0 if pre_lrelu is None else 1,
                                      C.c_float(0.0 if pre_lrelu is None else pre_lrelu), act,
                                      C.c_float(act_slope), _p(li, C.c_int32), _p(lo, C.c_int32)), "op_conv1d")
        return y

    def resblock_pair(self, x, w1, b1, w2, b2, dil=1, slope=0.1, fused=True, lens=None):
        x, w1, w2 = f32(x), f32(w1), f32(w2)
        B, Cc, T = x.shape
        K = w1.shape[2]
        y = np.empty_like(x)
        b1 = None if b1 is None else f32(b1)
        b2 = None if b2 is None else f32(b2)
        li = i32(lens)
        self._ck(lib().rvcx_op_resblock_pair(self._h, _p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(y), B, Cc, T, K, dil,
                                             C.c_float(slope), 1 if fused else 0, _p(li, C.c_int32)),
                 "op_resblock_pair")
        return y

    def resblock3(self, x, w1, b1, w2, b2, dils=(1, 3, 5), slope=0.1, lens=None):
        """a whole k = 3 ResBlock1 (three steps) in one kernel; w1 / w2 (3, C, C, 3), b1 / b2 (3, C) or None"""
        x, w1, w2 = f32(x), f32(w1), f32(w2)
        B, Cc, T = x.shape
        y = np.empty_like(x)
        b1 = None if b1 is None else f32(b1)
        b2 = None if b2 is None else f32(b2)
        d = i32(list(dils))
        li = i32(lens)
        self._ck(lib().rvcx_op_resblock3(self._h, _p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(y), B, Cc, T, _p(d, C.c_int32),
                                         C.c_float(slope), _p(li, C.c_int32)), "op_resblock3")
        return y

    def bench_resblock_pair(self, B, Cc, T, K, dil=1, fused=True, iters=10):
        ms = C.c_float(0)
        self._ck(lib().rvcx_bench_resblock_pair(self._h, B, Cc, T, K, dil, 1 if fused else 0, iters, C.byref(ms)),
                 "bench_resblock_pair")
        return ms.value, 4.0 * B * Cc * Cc * K * T / (ms.value * 1e-3) / 1e12

    @staticmethod
    def conv_override(tile=-1, variant=-1, splitk=-1):
        """tuning hook; the library refuses it (and debug_inject / bench_*) unless the process started with RVCX_DEBUG=1"""
        if lib().rvcx_conv_override(tile, variant, splitk) != 0:
            raise RvcxError("conv_override: " + (lib().rvcx_last_error(None) or b"").decode())

    def bench_conv1d(self, B, Cin, Tin, Cout, K, stride=1, dil=1, groups=1, iters=10):
        ms = C.c_float(0)
        self._ck(lib().rvcx_bench_conv1d(self._h, B, Cin, Tin, Cout, K, stride, dil, groups, iters, C.byref(ms)),
                 "bench_conv1d")
        pad = (K * dil - dil) // 2
        Tout = (Tin + 2 * pad - dil * (K - 1) - 1) // stride + 1
        flops = 2.0 * B * Cout * Tout * K * (Cin // groups)
        return ms.value, flops / (ms.value * 1e-3) / 1e12

    def convtranspose1d(self, x, w, bias=None, stride=1, pad=0, pre_lrelu=None):
        x, w = f32(x), f32(w)
        B, Cin, Tin = x.shape
        _, Cout, K = w.shape
        Tout = (Tin - 1) * stride - 2 * pad + K
        y = np.empty((B, Cout, Tout), np.float32)
        bias = None if bias is None else f32(bias)
        self._ck(lib().rvcx_op_convtranspose1d(self._h, _p(x), _p(w), _p(bias), _p(y), B, Cin, Tin, Cout, K,
                                               stride, pad, 0 if pre_lrelu is None else 1,
                                               C.c_float(0.0 if pre_lrelu is None else pre_lrelu)),
                 "op_convtranspose1d")
        return y

    def conv2d3x3(self, x, w, bias=None, res=None, act=0):
        x, w = f32(x), f32(w)
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        y = np.empty((B, Cout, H, W), np.float32)
        bias = None if bias is None else f32(bias)
        res = None if res is None else f32(res)
        self._ck(lib().rvcx_op_conv2d3x3(self._h, _p(x), _p(w), _p(bias), _p(res), _p(y), B, Cin, H, W, Cout,
                                         act), "op_conv2d3x3")
        return y

    def convblock2d(self, x, w1, b1, w2, b2, wsc=None, bsc=None, rows=None):
        """one ConvBlockRes (RMVPE.py:140-175, BN folded by the caller) through the F0 model's own block path"""
        x, w1, w2 = f32(x), f32(w1), f32(w2)
        B, Cin, H, W = x.shape
        Cout = w1.shape[0]
        y = np.empty((B, Cout, H, W), np.float32)
        b1 = None if b1 is None else f32(b1)
        b2 = None if b2 is None else f32(b2)
        wsc = None if wsc is None else f32(wsc)
        bsc = None if bsc is None else f32(bsc)
        ri = i32(rows)
        self._ck(lib().rvcx_op_convblock2d(self._h, _p(x), _p(w1), _p(b1), _p(w2), _p(b2), _p(wsc), _p(bsc), _p(y), B, Cin,
                                           Cout, H, W, _p(ri, C.c_int32)), "op_convblock2d")
        return y

    def convtranspose2d(self, x, w, bias=None, act=0):
        x, w = f32(x), f32(w)
        B, Cin, H, W = x.shape
        Cout = w.shape[1]
        y = np.empty((B, Cout, 2 * H, 2 * W), np.float32)
        bias = None if bias is None else f32(bias)
        self._ck(lib().rvcx_op_convtranspose2d(self._h, _p(x), _p(w), _p(bias), _p(y), B, Cin, H, W, Cout, act),
                 "op_convtranspose2d")
        return y

    def attention(self, q, k, v, heads, scale, emb_rel_k=None, emb_rel_v=None, window=10, lens=None):
        q, k, v = f32(q), f32(k), f32(v)
        B, HD, T = q.shape
        D = HD // heads
        out = np.empty_like(q)
        ek = None if emb_rel_k is None else f32(emb_rel_k)
        ev = None if emb_rel_v is None else f32(emb_rel_v)
        li = i32(lens)
        self._ck(lib().rvcx_op_attention(self._h, _p(q), _p(k), _p(v), _p(out), B, heads, D, T, C.c_float(scale),
                                         _p(ek), _p(ev), window, _p(li, C.c_int32)), "op_attention")
        return out

    def gemm_tm(self, x_cf, w, bias=None, res_tm=None, act=0, exact_fp32=False):
        """the time-major Linear kernel: x_cf (B, Cin, T) -> (y_tm (B*T, Cout), y_cf (B, Cout, T), y_split decoded)"""
        x_cf, w = f32(x_cf), f32(w)
        B, Cin, T = x_cf.shape
        Cout = w.shape[0]
        bias = None if bias is None else f32(bias)
        res = None if res_tm is None else f32(res_tm)
        y = np.empty((B * T, Cout), np.float32)
        ycf = np.empty((B, Cout, T), np.float32)
        ysp = np.empty((B * T, Cout), np.float32) if Cout % 16 == 0 else None
        self._ck(lib().rvcx_op_gemm_tm(self._h, _p(x_cf), _p(w), _p(bias), _p(res), B, T, Cin, Cout, int(act),
                                       1 if exact_fp32 else 0, _p(y), _p(ycf), _p(ysp)), "op_gemm_tm")
        return y, ycf, ysp

    def bench_gemm(self, rows, cin, cout, iters=20):
        ms = C.c_float(0)
        self._ck(lib().rvcx_bench_gemm(self._h, C.c_int64(rows), cin, cout, iters, C.byref(ms)), "bench_gemm")
        return ms.value, 2.0 * rows * cin * cout / (ms.value * 1e-3) / 1e12

    def layernorm_tm(self, x, gamma, beta, eps=1e-5):
        x = f32(x)
        rows, Cc = x.shape
        y = np.empty_like(x)
        ysp = np.empty_like(x) if Cc % 16 == 0 else None
        self._ck(lib().rvcx_op_layernorm_tm(self._h, _p(x), _p(f32(gamma)), _p(f32(beta)), _p(y), _p(ysp),
                                            C.c_int64(rows), Cc, C.c_float(eps)), "op_layernorm_tm")
        return y, ysp

    def layernorm_c(self, x, gamma, beta, eps=1e-5):
        x = f32(x)
        B, Cc, T = x.shape
        y = np.empty_like(x)
        self._ck(lib().rvcx_op_layernorm_c(self._h, _p(x), _p(f32(gamma)), _p(f32(beta)), _p(y), B, Cc, T,
                                           C.c_float(eps)), "op_layernorm_c")
        return y

    # ---- models -----------------------------------------------------------------------
    def load_synth(self, cfg_struct, state: dict) -> int:
        tbl, keep = make_table(state)
        mid = C.c_int(-1)
        self._ck(lib().rvcx_load_synth(self._h, C.byref(cfg_struct), tbl, len(tbl), C.byref(mid)), "load_synth")
        return mid.value

    def unload_synth(self, model_id: int):
        self._ck(lib().rvcx_unload_synth(self._h, int(model_id)), "unload_synth")

    def synth_upp(self, model_id: int) -> int:
        return int(lib().rvcx_synth_upp(self._h, model_id))

    def synth_dec_rf(self, model_id) -> int:
        """frames of z an output sample of the NSF decoder can depend on, each side (+ 2)"""
        return int(lib().rvcx_synth_dec_rf(self._h, int(model_id)))

    def synth_infer(self, model_id, phone, pitch, pitchf, lens=None, sid=None, z_noise=None, src_noise=None,
                    seed=0, taps=False, dec_skip=0):
        phone, pitchf = f32(phone), f32(pitchf)
        pitch = i32(pitch)
        B, T, _ = phone.shape
        upp = self.synth_upp(model_id)
        out = np.empty((B, T * upp), np.float32)
        lens = i32(np.full(B, T) if lens is None else lens)
        sid = i32(np.zeros(B) if sid is None else sid)
        zn = None if z_noise is None else f32(z_noise)
        sn = None if src_noise is None else f32(src_noise)
        if taps:
            inter = (zn.shape[1] if zn is not None else None)
            if inter is None:
                raise RvcxError("synth_infer(taps=True) needs z_noise (its shape gives inter_channels)")
            stats = np.empty((B, 2 * inter, T), np.float32)
            zflow = np.empty((B, inter, T), np.float32)
            self._ck(lib().rvcx_synth_infer_taps(self._h, model_id, B, T, _p(lens, C.c_int32), _p(phone),
                                                 _p(pitch, C.c_int32), _p(pitchf), _p(sid, C.c_int32), _p(zn), _p(sn),
                                                 C.c_uint64(seed), _p(out), _p(stats), _p(zflow)), "synth_infer_taps")
            return out, stats, zflow
        if dec_skip:
            self._ck(lib().rvcx_synth_infer_window(self._h, model_id, B, T, _p(lens, C.c_int32), _p(phone),
                                                   _p(pitch, C.c_int32), _p(pitchf), _p(sid, C.c_int32), _p(zn), _p(sn),
                                                   C.c_uint64(seed), int(dec_skip), _p(out)), "synth_infer_window")
            return out
        self._ck(lib().rvcx_synth_infer(self._h, model_id, B, T, _p(lens, C.c_int32), _p(phone),
                                        _p(pitch, C.c_int32), _p(pitchf), _p(sid, C.c_int32), _p(zn), _p(sn),
                                        C.c_uint64(seed), _p(out)), "synth_infer")
        return out

    def load_rmvpe(self, cfg_struct, state: dict):
        tbl, keep = make_table(state)
        self._ck(lib().rvcx_load_rmvpe(self._h, C.byref(cfg_struct), tbl, len(tbl)), "load_rmvpe")
        self.rmvpe_loaded = True

    def load_fcpe(self, cfg_struct, state: dict):
        tbl, keep = make_table(state)
        self._ck(lib().rvcx_load_fcpe(self._h, C.byref(cfg_struct), tbl, len(tbl)), "load_fcpe")
        self.fcpe_loaded = True

    def load_crepe(self, state: dict):
        """torchcrepe's model.Crepe state dict (capacity read off the shapes)."""
        tbl, keep = make_table(state)
        self._ck(lib().rvcx_load_crepe(self._h, tbl, len(tbl)), "load_crepe")
        self.crepe_loaded = True

    @staticmethod
    def crepe_frames(n: int, hop: int) -> int:
        return int(lib().rvcx_crepe_frames(C.c_int64(int(n)), int(hop)))

    def crepe_predict(self, x, hop, fmin, fmax, dither=None, seed=0, return_parts=False):
        """get_f0_crepe up to the resize: x / quantile -> torchcrepe.predict(..., batch_size=2*hop, pad=True): pitch (F,)
        [, sigmoid outputs (F, 360), Viterbi bins (F,)]."""
        x = f32(x)
        F = self.crepe_frames(x.shape[0], hop)
        pitch = np.empty(F, np.float32)
        probs = np.empty((360, F), np.float32) if return_parts else None
        bins = np.empty(F, np.int32) if return_parts else None
        d = None if dither is None else f32(dither)
        if d is not None and d.shape[0] < F:
            raise RvcxError("crepe dither shorter than the frame count")
        self._ck(lib().rvcx_crepe_predict(self._h, _p(x), C.c_int64(x.shape[0]), int(hop), C.c_float(fmin), C.c_float(fmax),
                                          _p(d), C.c_uint64(seed), _p(pitch), _p(probs), _p(bins, C.c_int32)), "crepe_predict")
        if return_parts:
            return pitch, np.ascontiguousarray(probs.T), bins
        return pitch

    def crepe_decode(self, probs, batch, fmin, fmax, dither):
        """core.postprocess + Viterbi + bins_to_frequency on sigmoid outputs (F, 360): (pitch (F,), bins (F,))."""
        pr = np.ascontiguousarray(f32(probs).T)
        F = pr.shape[1]
        d = f32(dither)
        pitch, bins = np.empty(F, np.float32), np.empty(F, np.int32)
        self._ck(lib().rvcx_op_crepe_decode(self._h, _p(pr), C.c_int64(F), int(batch), C.c_float(fmin), C.c_float(fmax),
                                            _p(d), _p(pitch), _p(bins, C.c_int32)), "crepe_decode")
        return pitch, bins

    def get_f0_crepe_x(self, x, p_len, params: "Params", inp_f0=None, dither=None):
        """VC.get_f0(..., "mangio-crepe", hop_length=params.hop_length) on the padded signal: (coarse, f0) of p_len frames."""
        x = f32(x)
        coarse, f0 = np.empty(int(p_len), np.int32), np.empty(int(p_len), np.float32)
        tab = None if inp_f0 is None else np.ascontiguousarray(inp_f0, dtype=np.float32).reshape(-1, 2)
        d = None if dither is None else f32(dither)
        self._ck(lib().rvcx_get_f0_crepe_x(self._h, _p(x), C.c_int64(x.shape[0]), C.c_int64(int(p_len)), C.byref(params),
                                           _p(tab), 0 if tab is None else tab.shape[0], _p(d),
                                           C.c_int64(0 if d is None else d.shape[0]), _p(coarse, C.c_int32), _p(f0)),
                 "get_f0_crepe_x")
        return coarse, f0

    def fcpe_f0(self, audio, threshold=0.05, return_salience=False, return_mel=False):
        """FCPEInfer.__call__: audio (n,) or (B,n) at 16 kHz -> Hz (B, n//160 + 1) [, salience (B,F,360)] [, mel (B,128,F)]."""
        audio = f32(audio)
        if audio.ndim == 1:
            audio = audio[None]
        B, n = audio.shape
        F = 1 + n // 160
        f0 = np.empty((B, F), np.float32)
        sal = np.empty((B, F, 360), np.float32) if return_salience else None
        mel = np.empty((B, 128, F), np.float32) if return_mel else None
        self._ck(lib().rvcx_fcpe_f0(self._h, B, _p(audio), C.c_int64(n), C.c_float(threshold), _p(f0), _p(sal),
                                    _p(mel)), "fcpe_f0")
        out = (f0,) + ((sal,) if return_salience else ()) + ((mel,) if return_mel else ())
        return out if len(out) > 1 else f0

    def get_f0_fcpe_x(self, x, p_len, params: "Params"):
        """VC.get_f0(..., f0_method="fcpe") on the already padded + filtered signal: (coarse, f0) of p_len frames."""
        x = f32(x)
        coarse = np.empty(int(p_len), np.int32)
        f0 = np.empty(int(p_len), np.float32)
        self._ck(lib().rvcx_get_f0_fcpe_x(self._h, _p(x), C.c_int64(x.shape[0]), C.c_int64(int(p_len)),
                                          C.byref(params), _p(coarse, C.c_int32), _p(f0)), "get_f0_fcpe_x")
        return coarse, f0

    def fcpe_post(self, raw, p_len, pitch=0.0, f0_min=50, f0_max=1100):
        raw = f32(raw)
        coarse = np.empty(int(p_len), np.int32)
        f0 = np.empty(int(p_len), np.float32)
        self._ck(lib().rvcx_op_fcpe_post(self._h, _p(raw), int(raw.shape[0]), int(p_len), C.c_double(pitch),
                                         C.c_double(f0_min), C.c_double(f0_max), _p(coarse, C.c_int32), _p(f0)),
                 "fcpe_post")
        return coarse, f0

    def load_hubert(self, cfg_struct, state: dict):
        tbl, keep = make_table(state)
        self._ck(lib().rvcx_load_hubert(self._h, C.byref(cfg_struct), tbl, len(tbl)), "load_hubert")

    def rmvpe_f0(self, audio, thred=0.03, f0_min=50, f0_max=1100, return_hidden=False):
        audio = f32(audio)
        if audio.ndim == 1:
            audio = audio[None]
        B, n = audio.shape
        F = 1 + n // 160
        f0 = np.empty((B, F), np.float32)
        hid = np.empty((B, F, 360), np.float32) if return_hidden else None
        self._ck(lib().rvcx_rmvpe_f0(self._h, B, _p(audio), C.c_int64(n), C.c_float(thred), C.c_float(f0_min),
                                     C.c_float(f0_max), _p(f0), _p(hid)), "rmvpe_f0")
        return (f0, hid) if return_hidden else f0

    def rmvpe_mel(self, audio):
        audio = f32(audio)
        if audio.ndim == 1:
            audio = audio[None]
        B, n = audio.shape
        mel = np.empty((B, 128, 1 + n // 160), np.float32)
        self._ck(lib().rvcx_rmvpe_mel(self._h, B, _p(audio), C.c_int64(n), _p(mel)), "rmvpe_mel")
        return mel

    def hubert_frames(self, n: int) -> int:
        return int(lib().rvcx_hubert_frames(self._h, C.c_int64(n)))

    def hubert_features(self, wav, embed_dim, output_layer=12):
        wav = f32(wav)
        if wav.ndim == 1:
            wav = wav[None]
        B, n = wav.shape
        T = self.hubert_frames(n)
        out = np.empty((B, T, embed_dim), np.float32)
        self._ck(lib().rvcx_hubert_features(self._h, B, _p(wav), C.c_int64(n), output_layer, _p(out)),
                 "hubert_features")
        return out

    def bigru(self, x, sd, prefix="fc.0.gru"):
        x = f32(x)
        B, T, I = x.shape
        g = lambda n: f32(sd[f"{prefix}.{n}"])
        H = g("weight_hh_l0").shape[1]
        y = np.empty((B, T, 2 * H), np.float32)
        a = [g("weight_ih_l0"), g("weight_hh_l0"), g("bias_ih_l0"), g("bias_hh_l0"), g("weight_ih_l0_reverse"),
             g("weight_hh_l0_reverse"), g("bias_ih_l0_reverse"), g("bias_hh_l0_reverse")]
        self._ck(lib().rvcx_op_bigru(self._h, _p(x), *[_p(t) for t in a], _p(y), B, T, I, H), "op_bigru")
        return y

    def load_index(self, big_npy):
        if big_npy is None:
            self._ck(lib().rvcx_load_index(self._h, None, C.c_int64(0), 0), "load_index")
            return
        b = f32(big_npy)
        self._ck(lib().rvcx_load_index(self._h, _p(b), C.c_int64(b.shape[0]), int(b.shape[1])), "load_index")

    def load_index_ivf(self, big_npy, centroids, assign, nprobe=1):
        """faiss "IVF{nlist},Flat": stored vectors + coarse centroids + list id per vector; nprobe = 1 search."""
        b, c = f32(big_npy), f32(centroids)
        a = i32(assign)
        self._ck(lib().rvcx_load_index_ivf(self._h, _p(b), C.c_int64(b.shape[0]), int(b.shape[1]), _p(c), int(c.shape[0]),
                                           _p(a, C.c_int32), int(nprobe)), "load_index_ivf")

    def index_blend(self, feats, index_rate):
        f = f32(feats).copy()
        T = f.shape[0]
        ids = np.empty((T, 8), np.int64)
        dist = np.empty((T, 8), np.float32)
        self._ck(lib().rvcx_index_blend(self._h, _p(f), T, C.c_float(index_rate), _p(ids, C.c_int64), _p(dist)),
                 "index_blend")
        return f, ids, dist

    def highpass(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self._ck(lib().rvcx_op_highpass(self._h, _p(x, C.c_double), _p(y, C.c_double), C.c_int64(x.shape[0])),
                 "op_highpass")
        return y

    def get_f0(self, wav, params: "Params"):
        wav = f32(wav)
        n = wav.shape[0]
        cap = (n + 2 * 16000 * params.x_pad) // 160 + 8
        coarse = np.empty(cap, np.int32)
        f0 = np.empty(cap, np.float32)
        pl = C.c_int64(0)
        self._ck(lib().rvcx_get_f0(self._h, _p(wav), C.c_int64(n), C.byref(params), _p(coarse, C.c_int32), _p(f0),
                                   C.byref(pl)), "get_f0")
        return coarse[:pl.value].copy(), f0[:pl.value].copy()

    def out_capacity(self, model_id, n, params) -> int:
        return int(lib().rvcx_out_len(self._h, model_id, C.c_int64(n), C.byref(params)))

    def noise_capacity(self, model_id, n, params) -> int:
        return int(lib().rvcx_noise_len(self._h, model_id, C.c_int64(n), C.byref(params)))

    def convert_batch(self, model_id, wavs, params: "Params", noises=None, want_f32=False, inp_f0=None, crepe_dither=None):
        """VC.pipeline for a list of 16 kHz mono clips -> list of int16 arrays (and the pre-quantisation
        float waveforms when want_f32).  float64 clips (what the reference's load_audio returns) cross the
        ABI as float64; anything else as float32.  Clips of one length class are converted as ragged micro-batches."""
        B = len(wavs)
        is64 = B > 0 and all(np.asarray(w).dtype == np.float64 for w in wavs)
        wavs = [np.ascontiguousarray(w, dtype=np.float64 if is64 else np.float32) for w in wavs]
        ns = (C.c_int64 * B)(*[w.shape[0] for w in wavs])
        wt = C.c_double if is64 else C.c_float
        wp = (C.POINTER(wt) * B)(*[_p(w, wt) for w in wavs])
        caps = [self.out_capacity(model_id, w.shape[0], params) for w in wavs]
        if any(c < 0 for c in caps):
            raise RvcxError(f"convert_batch: no voice model with id {model_id} is resident in this context")
        outs = [np.empty(c, np.int16) for c in caps]
        op = (C.POINTER(C.c_int16) * B)(*[_p(o, C.c_int16) for o in outs])
        f32s, fp = None, None
        if want_f32:
            f32s = [np.empty(c, np.float32) for c in caps]
            fp = (C.POINTER(C.c_float) * B)(*[_p(o) for o in f32s])
        nz, npp = None, None
        if noises is not None:
            nz = []
            for w, nv in zip(wavs, noises):
                if nv is None:
                    nz.append(None)
                    continue
                cap = self.noise_capacity(model_id, w.shape[0], params)
                buf = np.zeros(cap, np.float32)
                nv = f32(nv).ravel()
                if nv.shape[0] > cap:
                    raise RvcxError("noise longer than rvcx_noise_len")
                buf[:nv.shape[0]] = nv
                nz.append(buf)
            npp = (C.POINTER(C.c_float) * B)(*[_p(b) for b in nz])
        out_n = (C.c_int64 * B)()
        if inp_f0 is not None or crepe_dither is not None:
            # f0 files: (rows, 2) float32 tables of (time [s], f0 [Hz]) per utterance; crepe dither: one float per frame
            tabs = [None if (inp_f0 is None or t is None) else np.ascontiguousarray(t, dtype=np.float32).reshape(-1, 2)
                    for t in (inp_f0 if inp_f0 is not None else [None] * B)]
            dith = [None if (crepe_dither is None or d is None) else f32(d).ravel()
                    for d in (crepe_dither if crepe_dither is not None else [None] * B)]
            ex = (UttExtra * B)(*[UttExtra(_p(t), 0 if t is None else t.shape[0], 0, _p(d), 0 if d is None else d.shape[0])
                                  for t, d in zip(tabs, dith)])
            wv = (C.c_void_p * B)(*[w.ctypes.data for w in wavs])
            self._ck(lib().rvcx_convert_batch_ex(self._h, model_id, B, wv, 1 if is64 else 0, ns, C.byref(params), npp, ex,
                                                 op, fp, out_n), "convert_batch_ex")
        else:
            fn = lib().rvcx_convert_batch_f64 if is64 else lib().rvcx_convert_batch
            self._ck(fn(self._h, model_id, B, wp, ns, C.byref(params), npp, op, fp, out_n), "convert_batch")
        pcm = [o[:out_n[i]].copy() for i, o in enumerate(outs)]
        if want_f32:
            return pcm, [o[:out_n[i]].copy() for i, o in enumerate(f32s)]
        return pcm

    def convert_batch_raw(self, model_id, wav_ptrs, ns, params, out_ptrs, f32_ptrs=None, noise_ptrs=None):
        """Same as convert_batch but with raw (host or device) addresses of float32 clips: nothing is staged
        through numpy.  Returns the list of produced sample counts."""
        B = len(wav_ptrs)
        wp = (C.c_void_p * B)(*wav_ptrs)
        op = (C.c_void_p * B)(*out_ptrs)
        fp = None if f32_ptrs is None else (C.c_void_p * B)(*f32_ptrs)
        npp = None if noise_ptrs is None else (C.c_void_p * B)(*noise_ptrs)
        nn = (C.c_int64 * B)(*ns)
        out_n = (C.c_int64 * B)()
        self._ck(lib().rvcx_convert_batch(self._h, model_id, B, wp, nn, C.byref(params), npp, op, fp, out_n),
                 "convert_batch")
        return [int(v) for v in out_n]

    def micro_batch(self, model_id, n, params) -> int:
        return int(lib().rvcx_micro_batch(self._h, model_id, C.c_int64(n), C.byref(params)))

    def bucket_length(self, model_id, n, params) -> int:
        """the length whose launch geometry an n-sample utterance runs with: equal values share micro-batches"""
        return int(lib().rvcx_bucket_length(self._h, model_id, C.c_int64(n), C.byref(params)))

    def last_micro_batches(self):
        """member counts of the micro-batches the last convert_batch call formed"""
        cap = 4096
        buf = (C.c_int32 * cap)()
        k = int(lib().rvcx_last_micro_batches(self._h, buf, cap))
        return [int(buf[i]) for i in range(min(k, cap))]

    def get_f0_x(self, x, params: "Params"):
        """VC.get_f0 on the already padded + filtered signal: (coarse, f0) of 1 + n/160 frames."""
        x = f32(x)
        F = 1 + x.shape[0] // 160
        coarse = np.empty(F, np.int32)
        f0 = np.empty(F, np.float32)
        self._ck(lib().rvcx_get_f0_x(self._h, _p(x), C.c_int64(x.shape[0]), C.byref(params), _p(coarse, C.c_int32),
                                     _p(f0)), "get_f0_x")
        return coarse, f0

    def get_f0_x_ex(self, x, p_len, params: "Params", inp_f0=None):
        """VC.get_f0 in full (F0 model by params.f0_method, pitch shift, optional f0-file table, coarse) on the padded
        signal: (coarse, f0) of 1 + n/160 frames (rmvpe+) or p_len frames (fcpe)."""
        x = f32(x)
        cap = max(int(p_len), 1 + x.shape[0] // 160)
        coarse = np.empty(cap, np.int32)
        f0 = np.empty(cap, np.float32)
        tab = None if inp_f0 is None else np.ascontiguousarray(inp_f0, dtype=np.float32).reshape(-1, 2)
        got = C.c_int64(0)
        self._ck(lib().rvcx_get_f0_x_ex(self._h, _p(x), C.c_int64(x.shape[0]), C.c_int64(int(p_len)), C.byref(params),
                                        _p(tab), 0 if tab is None else tab.shape[0], _p(coarse, C.c_int32), _p(f0),
                                        C.byref(got)), "get_f0_x_ex")
        return coarse[:got.value].copy(), f0[:got.value].copy()

    def resample(self, audio, sr_in: int, sr_out: int, kind: int = -1) -> np.ndarray:
        """librosa.resample(librosa.to_mono(audio.T), sr_in -> sr_out): audio (frames,) or (frames, channels) -> float64 mono.
        kind -1: the default filter (kaiser_hq: a Kaiser design to soxr_hq's published targets), 1: resampy's kaiser_best"""
        a = np.ascontiguousarray(audio, dtype=np.float64)
        frames, ch = a.shape[0], (1 if a.ndim == 1 else a.shape[1])
        n_out = int(lib().rvcx_resample_len(C.c_int64(frames), int(sr_in), int(sr_out)))
        y = np.empty(max(n_out, 0), np.float64)
        self._ck(lib().rvcx_resample_f64_kind(self._h, _p(a, C.c_double), C.c_int64(frames), ch, int(sr_in), int(sr_out),
                                              int(kind), _p(y, C.c_double)), "resample_f64")
        return y

    def vc_frames(self, n: int) -> int:
        return int(lib().rvcx_vc_frames(self._h, C.c_int64(n)))

    def vc(self, model_id, audio0, pitch, pitchf, sid=0, index_rate=0.0, protect=0.5, z_noise=None, src_noise=None,
           seed=0):
        """VC.vc: one chunk of audio_pad -> the un-trimmed float32 waveform."""
        a = f32(audio0)
        pitch, pitchf = i32(np.asarray(pitch).ravel()), f32(np.asarray(pitchf).ravel())
        T = self.vc_frames(a.shape[0])
        if T <= 0:
            raise RvcxError("vc: chunk too short")
        out = np.empty(T * self.synth_upp(model_id), np.float32)
        zn = None if z_noise is None else f32(z_noise)
        sn = None if src_noise is None else f32(src_noise)
        got = C.c_int64(0)
        self._ck(lib().rvcx_vc(self._h, model_id, _p(a), C.c_int64(a.shape[0]), _p(pitch, C.c_int32), _p(pitchf),
                               int(min(pitch.shape[0], pitchf.shape[0])), int(sid), C.c_float(index_rate),
                               C.c_float(protect), _p(zn), _p(sn), C.c_uint64(seed), _p(out), C.byref(got)), "vc")
        return out[:got.value]

    def weights_regions(self):
        """[(device pointer, used bytes)] of every weight chunk + the shape-only layout hash."""
        h = C.c_uint64(0)
        n = lib().rvcx_weights_regions(self._h, 0, None, None, C.byref(h))
        if n < 0:
            self._ck(-1, "weights_regions")
        ptrs, sizes = (C.c_void_p * max(n, 1))(), (C.c_int64 * max(n, 1))()
        n2 = lib().rvcx_weights_regions(self._h, n, ptrs, sizes, C.byref(h))
        if n2 != n:
            raise RvcxError("weights_regions: chunk count changed between calls")
        return [(int(ptrs[i] or 0), int(sizes[i])) for i in range(n)], int(h.value)

    def weights_clone(self, src: "Context"):
        """copy the folded weights of ``src`` (same GPU, same model configurations) into this context"""
        self._ck(lib().rvcx_weights_clone(self._h, src._h), "weights_clone")

    def weights_adopt(self):
        self._ck(lib().rvcx_weights_adopt(self._h), "weights_adopt")

    def conv_profile_begin(self):
        z = (C.c_int64 * 72)()
        self._ck(lib().rvcx_conv_profile(self._h, 1, z, None, None, None, None, None, 0), "conv_profile")

    def conv_profile_end(self):
        N = 72
        la, fl, ms = (C.c_int64 * N)(), (C.c_double * N)(), (C.c_double * N)()
        bm, bn, kd = (C.c_int32 * N)(), (C.c_int32 * N)(), (C.c_int32 * N)()
        self._ck(lib().rvcx_conv_profile(self._h, 0, la, fl, ms, bm, bn, kd, N), "conv_profile")

        def name(i):
            if kd[i] < 0:
                return f"conv_mfma_kernel<{bm[i]},{bn[i]}> (generic, strided/grouped)"
            if kd[i] == 600001:
                return "gemm_f32<64,64> (time-major Linear, exact fp32)"
            if kd[i] == 600002:
                return f"gemm_bd<{bm[i]},{bn[i]}> (time-major Linear, activations straight from global memory)"
            if kd[i] >= 600000:
                return f"gemm_h3<{bm[i]},{bn[i]}> (time-major Linear)"
            if kd[i] == 500003:
                return "resblock3 (whole k=3 ResBlock: three fused steps, persistent)"
            if kd[i] >= 500000:
                return f"resblock_pair<C={kd[i] - 500000},N1={bn[i]}> (c1 -> c2 -> +x fused)"
            if kd[i] >= 400000:
                c = kd[i] - 400000
                return f"conv_h3<{bm[i]},{bn[i]},{'linear' if c == 1 else 'stride2' if c == 2 else 'halo%d' % c}>"
            if kd[i] == 300001:
                return "conv_cout1 (vector FMA, Cout=1)"
            if kd[i] == 300003:
                return "conv_ws<64,320> (weight-stationary tile: a stage = one 16-channel chunk x all taps; K segments + finish)"
            if kd[i] == 300004:
                return "conv3_thin (3x3, C=16/32: streaming MFMA, B operand from global, DPP-shifted taps)"
            if kd[i] == 300002:
                return "convt_thin (ConvTranspose1d k4 s2, streaming MFMA + fused noise conv)"
            if kd[i] >= 300000:
                return "conv_cin1 (vector FMA, Cin=1)"
            if kd[i] >= 200000:
                return f"conv_fast_sb<{bm[i]},{bn[i]},stride2>"
            if kd[i] >= 100000:
                return f"conv_fast_sb<{bm[i]},{bn[i]},linear>"
            return f"conv_fast_sb<{bm[i]},{bn[i]},halo{kd[i] // 10}>"
        self._tile_names = [name(i) for i in range(N)]
        return [dict(tile=name(i), launches=int(la[i]), flops=float(fl[i]), ms=float(ms[i]))
                for i in range(N) if la[i] > 0]

    def conv_profile_csv(self) -> str:
        """per-launch table of the last profile; the tile column carries the kernel's name (conv_profile_end first)"""
        txt = (lib().rvcx_conv_profile_csv(self._h) or b"").decode()
        names = getattr(self, "_tile_names", None)
        if not names:
            return txt
        out = []
        for i, line in enumerate(txt.splitlines()):
            head, _, rest = line.partition(",")
            if i > 0 and head.isdigit() and int(head) < len(names):
                head = '"' + names[int(head)].split(" (")[0] + '"'
            out.append(head + "," + rest)
        return "\n".join(out) + "\n"

    def last_timing(self):
        ms = (C.c_float * 9)()
        lib().rvcx_last_timing(self._h, ms)
        names = ["highpass", "rmvpe", "hubert", "index", "enc_p", "flow", "decoder", "post", "total"]
        return dict(zip(names, [float(v) for v in ms]))

    def mem_info(self):
        f, t = C.c_int64(0), C.c_int64(0)
        self._ck(lib().rvcx_mem_info(self._h, C.byref(f), C.byref(t)), "mem_info")
        return f.value, t.value

    def fp32_reruns(self) -> int:
        """calls repeated on the exact-fp32 kernels after an fp16-split overflow"""
        return int(lib().rvcx_fp32_reruns(self._h))

    def fp32_pinned(self) -> str:
        """text, one line per layer the range guard pinned to the exact-fp32 kernels at run time (rvcx_fp32_pinned)"""
        buf = C.create_string_buffer(1 << 16)
        if lib().rvcx_fp32_pinned(self._h, buf, len(buf)) < 0:
            raise RvcxError("fp32_pinned")
        return buf.value.decode()

    def fp32_layers(self) -> int:
        """layers pinned to the exact-fp32 kernels after an activation left fp16 range (sticky per model)"""
        return int(lib().rvcx_fp32_layers(self._h))

    def index_exhaustive(self) -> int:
        """queries searched exhaustively (pre-filter not certifiable) since the last call; clears the counter"""
        return int(lib().rvcx_index_exhaustive(self._h))

    def gru_fallbacks(self) -> int:
        return int(lib().rvcx_gru_fallbacks(self._h))

    def gru_publish_probe(self) -> int:
        """1: the cluster BiGRU's plain-store publish was verified on this device, 0: it failed and the write-through
        publish is used, -1: undecided"""
        return int(lib().rvcx_gru_publish_probe(self._h))

    def debug_inject(self, what: int):
        self._ck(lib().rvcx_debug_inject(self._h, int(what)), "debug_inject")

    def flop_counter(self, reset=False) -> float:
        return float(lib().rvcx_flop_counter(self._h, 1 if reset else 0))
