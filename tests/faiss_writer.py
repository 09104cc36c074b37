"""Test-side ENCODER of the faiss 1.7 index file layouts rvcx reads (polgen-rvc_amd/index_io.py), written from
the published io code (faiss/impl/index_write.cpp: write_index_header, write_ivf_header, write_direct_map,
write_InvertedLists).  faiss is not installed here; these writers let the parser and the IVF search be tested on
byte streams of the real layout.  Never imported by the product."""
import struct

import numpy as np


def _header(d, ntotal, metric=1):
    return struct.pack("<iqqqBi", d, ntotal, 1 << 20, 1 << 20, 1, metric)


def _vec(a, fmt):
    a = np.ascontiguousarray(a, fmt)
    return struct.pack("<Q", a.size) + a.tobytes()


def flat_bytes(vectors) -> bytes:
    v = np.ascontiguousarray(vectors, np.float32)
    return b"IxF2" + _header(v.shape[1], v.shape[0]) + _vec(v.ravel(), np.float32)


def ivf_flat_bytes(vectors, centroids, assign, nprobe=1, sparse_sizes=False) -> bytes:
    """IndexIVFFlat with array inverted lists; ids are the row numbers, as ``index.add(big_npy)`` assigns them."""
    v = np.ascontiguousarray(vectors, np.float32)
    c = np.ascontiguousarray(centroids, np.float32)
    n, d = v.shape
    nlist = c.shape[0]
    out = b"IwFl" + _header(d, n) + struct.pack("<QQ", nlist, nprobe) + flat_bytes(c)
    out += struct.pack("<B", 0) + struct.pack("<Q", 0)                  # direct map: NoMap, empty array
    out += b"ilar" + struct.pack("<QQ", nlist, 4 * d)
    lists = [np.where(np.asarray(assign) == li)[0].astype(np.int64) for li in range(nlist)]
    sizes = np.array([len(l) for l in lists], np.uint64)
    if sparse_sizes:
        nz = np.nonzero(sizes)[0]
        pairs = np.stack([nz.astype(np.uint64), sizes[nz]], axis=1).ravel()
        out += b"sprs" + _vec(pairs, np.uint64)
    else:
        out += b"full" + _vec(sizes, np.uint64)
    for ids in lists:
        if len(ids):
            out += v[ids].tobytes() + ids.tobytes()
    return out
