"""Process-wide state shared by the two mirror modules: one rvcx context per GPU, the resident-asset
tables, and the Philox seed source for production noise."""
from __future__ import annotations

import itertools
import os

from .. import _lib

_CTX = {}                 # device index -> _lib.Context
_RESIDENT = {}            # (device index, kind) -> (file key, handle)      kind in {"hubert", "rmvpe"}
_SYNTHS = {}              # (device index, file key) -> (light cpt, SynthHandle); insertion order = LRU order
_INDEX_RESIDENT = {}      # id(context) -> key of the FAISS vectors resident in HBM (file key or ("array", id))

# The reference draws its two Gaussian tensors per chunk from torch's global generator
# (synthesizers.py:174, generators.py:154): every request gets fresh noise.  Here every call takes the next
# seed of a process-wide sequence that starts at a random 64-bit value (VC.seed pins it for reproducible runs).
_seed_base = int.from_bytes(os.urandom(8), "little")
_seed_counter = itertools.count()


def next_seed() -> int:
    return (_seed_base + 0x9E3779B97F4A7C15 * next(_seed_counter)) & 0xFFFFFFFFFFFFFFFF


def dev_index(device) -> int:
    if isinstance(device, str) and ":" in device:
        return int(device.split(":")[1])
    return device if isinstance(device, int) else 0


def default_device() -> int:
    """The GPU of calls that name no device (load_audio): the one this process already serves -- its resident context,
    else a torchrun-style LOCAL_RANK, else 0.  A rank on cuda:N must not build a second full context on GPU 0."""
    if _CTX:
        return next(iter(_CTX))
    return int(os.environ.get("LOCAL_RANK", "0") or 0)


def context(device=None) -> "_lib.Context":
    """One rvcx context per GPU, created on first use."""
    idx = default_device() if device is None else dev_index(device)
    if idx not in _CTX:
        _CTX[idx] = _lib.Context(idx)
    return _CTX[idx]


def file_key(path):
    st = os.stat(path)
    return (os.path.realpath(path), st.st_mtime_ns, st.st_size)
