"""``.index`` files -> the (N, dim) float32 matrix ``index.reconstruct_n(0, ntotal)`` returns
(rvc/infer/pipeline.py:322-323), without faiss.

Supported: ``.npy`` dumps of big_npy, FAISS ``IndexFlat`` ("IxF2"/"IxFI") and ``IndexIVFFlat`` ("IwFl",
array inverted lists).  The FAISS binary layouts are restated from the published faiss 1.7 io format;
faiss is not installed here, so the FAISS branches are UNVERIFIED against a real file (parity unpinned).
Search in rvcx is exact brute force; IVF probing (nprobe) is not reproduced.
"""
from __future__ import annotations

import struct

import numpy as np


class _R:
    def __init__(self, b):
        self.b, self.o = b, 0

    def take(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.o)
        self.o += struct.calcsize("<" + fmt)
        return v if len(v) > 1 else v[0]

    def fourcc(self):
        s = self.b[self.o:self.o + 4].decode("latin1")
        self.o += 4
        return s

    def vec(self, dtype):
        n = self.take("Q")
        a = np.frombuffer(self.b, dtype=dtype, count=n, offset=self.o)
        self.o += a.nbytes
        return a


def _header(r):
    d = r.take("i")
    ntotal = r.take("q")
    r.take("q")
    r.take("q")
    r.take("B")      # is_trained
    r.take("i")      # metric_type
    return d, ntotal


def _read_index(r):
    cc = r.fourcc()
    if cc in ("IxF2", "IxFI", "IxFl"):
        d, ntotal = _header(r)
        xb = r.vec(np.float32)
        return xb.reshape(ntotal, d).copy(), None
    if cc == "IwFl":
        d, ntotal = _header(r)
        nlist, nprobe = r.take("Q"), r.take("Q")
        _read_index(r)                       # coarse quantizer
        dm_type = r.take("B")                # direct map
        r.vec(np.int64)
        if dm_type == 2:
            raise ValueError("hashtable direct map not supported")
        il = r.fourcc()
        if il != "ilar":
            raise ValueError(f"unsupported inverted lists {il!r}")
        nl, code_size = r.take("Q"), r.take("Q")
        lt = r.fourcc()
        if lt == "full":
            sizes = r.vec(np.uint64)
        elif lt == "sprs":
            sp = r.vec(np.uint64)
            sizes = np.zeros(nl, np.uint64)
            sizes[sp[0::2].astype(np.int64)] = sp[1::2]
        else:
            raise ValueError(f"unsupported list type {lt!r}")
        out = np.zeros((ntotal, d), np.float32)
        for sz in sizes:
            sz = int(sz)
            codes = np.frombuffer(r.b, np.uint8, sz * code_size, r.o)
            r.o += sz * code_size
            ids = np.frombuffer(r.b, np.int64, sz, r.o)
            r.o += sz * 8
            if sz:
                out[ids] = codes.view(np.float32).reshape(sz, d)
        return out, None
    raise ValueError(f"unsupported FAISS index type {cc!r}")


def read_index_vectors(path: str) -> np.ndarray:
    if path.endswith(".npy"):
        a = np.load(path)
        if a.ndim != 2:
            raise ValueError("big_npy must be 2-D")
        return np.ascontiguousarray(a, np.float32)
    with open(path, "rb") as f:
        data = f.read()
    mat, _ = _read_index(_R(data))
    return mat
