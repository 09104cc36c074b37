"""Process-wide state shared by the two mirror modules: one rvcx context per GPU, the resident-asset
tables, and the Philox seed source for production noise."""
from __future__ import annotations

import functools
import itertools
import os
import threading

from .. import _lib

# The reference's callers run on Gradio worker threads and build fresh objects per request
# (rvc/scripts/voice_conversion.py:71-100), so concurrent requests are safe there.  Here the threads of a process share
# one context and the resident-asset tables below: LOCK guards the tables (context creation, load_*, get_vc), and every
# mirror method that talks to a context holds that context's own lock for its whole multi-call sequence
# (`with_ctx_lock`).  Concurrent requests therefore queue; batches (VC.pipeline_batch) are where throughput comes from.
LOCK = threading.RLock()


def locked(fn):
    """run `fn` under the process-wide table lock"""
    @functools.wraps(fn)
    def wrapper(*a, **k):
        with LOCK:
            return fn(*a, **k)
    return wrapper


def locked_dev(fn):
    """for loaders `fn(device, ...)`: hold the device's context lock, THEN the table lock -- the same order a mirror
    method uses when it loads a predictor lazily in the middle of a request (never the reverse: no deadlock), and a
    model is never swapped under another thread's half-finished request"""
    @functools.wraps(fn)
    def wrapper(device, *a, **k):
        with context(device).lock:
            with LOCK:
                return fn(device, *a, **k)
    return wrapper


_CTX = {}                 # device index -> _lib.Context
_RESIDENT = {}            # (device index, kind) -> (file key, handle)      kind in {"hubert", "rmvpe"}
_SYNTHS = {}              # (device index, file key) -> (light cpt, SynthHandle); insertion order = LRU order
_INDEX_RESIDENT = {}      # id(context) -> key of the FAISS vectors resident in HBM (file key or ("array", id))

# The reference draws its two Gaussian tensors per chunk from torch's global generator
# (synthesizers.py:174, generators.py:154): every request gets fresh noise.  Here every call takes the next
# seed of a process-wide sequence that starts at a random 64-bit value (VC.seed pins it for reproducible runs).
_seed_base = int.from_bytes(os.urandom(8), "little")
_seed_counter = itertools.count()      # next() on it is atomic under the GIL


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
    """One rvcx context per GPU, created on first use.  The hit path is a plain dict read (atomic under the GIL): a
    request never waits here for another thread's model load; LOCK is taken only to create."""
    idx = default_device() if device is None else dev_index(device)
    c = _CTX.get(idx)
    if c is not None:
        return c
    with LOCK:
        if idx not in _CTX:
            _CTX[idx] = _lib.Context(idx)
        return _CTX[idx]


def file_key(path):
    st = os.stat(path)
    return (os.path.realpath(path), st.st_mtime_ns, st.st_size)
