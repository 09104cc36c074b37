"""``.index`` files -> the (N, dim) float32 matrix ``index.reconstruct_n(0, ntotal)`` returns
(rvc/infer/pipeline.py:322-323), without faiss.

Supported: ``.npy`` dumps of big_npy, FAISS ``IndexFlat`` ("IxF2"/"IxFI") and ``IndexIVFFlat`` ("IwFl",
array inverted lists).  The FAISS binary layouts are restated from the published faiss 1.7 io format
(impl/index_read.cpp: read_index_header, read_ivf_header, read_direct_map, read_InvertedLists); faiss is not
installed here, so they are checked against a test-side encoder of the same layout (tests/faiss_writer.py), not
against a file faiss wrote (parity unpinned).  For an IVF file the coarse centroids, the list of every stored
vector and ``nprobe`` are returned too: rvcx then searches like faiss does -- only the query's nearest list when
nprobe = 1 (what RVC index files carry).
"""
from __future__ import annotations

import struct

import numpy as np


class UnsupportedIndex(ValueError):
    """A well-formed FAISS file of a kind (or with an option) rvcx does not implement.  faiss would read and search
    it, so the caller must hear about it -- unlike a corrupt file, which the reference swallows too."""


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


class IndexFile:
    """What ``faiss.read_index`` + ``reconstruct_n(0, ntotal)`` give the reference (pipeline.py:322-323)."""

    def __init__(self, vectors, centroids=None, assign=None, nprobe=None):
        self.vectors, self.centroids, self.assign, self.nprobe = vectors, centroids, assign, nprobe

    @property
    def is_ivf(self):
        return self.centroids is not None


def _read_index(r):
    cc = r.fourcc()
    if cc in ("IxF2", "IxFI", "IxFl"):
        d, ntotal = _header(r)
        xb = r.vec(np.float32)
        return xb.reshape(ntotal, d).copy(), None
    if cc == "IwFl":
        d, ntotal = _header(r)
        nlist, nprobe = r.take("Q"), r.take("Q")
        centroids, _ = _read_index(r)        # coarse quantizer (an IndexFlat of nlist centroids)
        if centroids.shape != (nlist, d):
            raise ValueError("IVF coarse quantizer does not hold nlist centroids")
        dm_type = r.take("B")                # direct map
        r.vec(np.int64)
        if dm_type == 2:
            raise UnsupportedIndex("hashtable direct map not supported")
        il = r.fourcc()
        if il != "ilar":
            raise UnsupportedIndex(f"unsupported inverted lists {il!r}")
        nl, code_size = r.take("Q"), r.take("Q")
        lt = r.fourcc()
        if lt == "full":
            sizes = r.vec(np.uint64)
        elif lt == "sprs":
            sp = r.vec(np.uint64)
            sizes = np.zeros(nl, np.uint64)
            sizes[sp[0::2].astype(np.int64)] = sp[1::2]
        else:
            raise UnsupportedIndex(f"unsupported list type {lt!r}")
        if code_size != 4 * d or nl != nlist:
            raise UnsupportedIndex("IVF inverted lists are not flat float32 codes")
        out = np.zeros((ntotal, d), np.float32)
        assign = np.full(ntotal, -1, np.int32)
        for li, sz in enumerate(sizes):
            sz = int(sz)
            codes = np.frombuffer(r.b, np.uint8, sz * code_size, r.o)
            r.o += sz * code_size
            ids = np.frombuffer(r.b, np.int64, sz, r.o)
            r.o += sz * 8
            if sz:
                out[ids] = codes.view(np.float32).reshape(sz, d)
                assign[ids] = li
        if (assign < 0).any():
            raise ValueError("IVF index: stored ids are not 0..ntotal-1 (reconstruct_n would fail in the reference too)")
        return out, dict(centroids=centroids, assign=assign, nprobe=int(nprobe))
    if cc[:1] == "I" and cc.isprintable():          # some other faiss index class ("IwPQ", "IxPq", "IHNf", ...)
        raise UnsupportedIndex(f"unsupported FAISS index type {cc!r} (rvcx reads IndexFlat and IndexIVFFlat files)")
    raise ValueError(f"not a FAISS index file (starts with {cc!r})")


def read_index(path: str) -> IndexFile:
    if path.endswith(".npy"):
        a = np.load(path)
        if a.ndim != 2:
            raise ValueError("big_npy must be 2-D")
        return IndexFile(np.ascontiguousarray(a, np.float32))
    with open(path, "rb") as f:
        data = f.read()
    mat, ivf = _read_index(_R(data))
    if ivf is None:
        return IndexFile(mat)
    return IndexFile(mat, np.ascontiguousarray(ivf["centroids"], np.float32), ivf["assign"], ivf["nprobe"])


def read_index_vectors(path: str) -> np.ndarray:
    return read_index(path).vectors
