"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" in the CPU tests).  The conversion path shards by utterance -- there is NO collective in
the hot loop.  The only collective is the one-off broadcast of the folded weight slab
from rank 0, so checkpoints are parsed/folded/packed once per node instead of once per GPU.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Sequence

import torch
import torch.distributed as dist


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend: str = "nccl"):
    rank, local, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard(n_items: int, rank: int, world: int, lengths: Sequence[int] = None) -> List[int]:
    """Static partition of utterances over ranks: length-sorted round-robin (longest first) so that every
    rank gets the same number of items (+-1) and a similar amount of audio."""
    order = list(range(n_items))
    if lengths is not None:
        order.sort(key=lambda i: (-lengths[i], i))
    return sorted(order[rank::world])


def broadcast_tensor(t: torch.Tensor, src: int = 0) -> torch.Tensor:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src)
    return t


class _DevBytes:
    """zero-copy uint8 view of raw device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def _view(ptr: int, nbytes: int, dev) -> torch.Tensor:
    """uint8 tensor aliasing [ptr, ptr + nbytes): device memory under nccl, host memory under gloo."""
    if dev.type == "cuda":
        return torch.as_tensor(_DevBytes(ptr, nbytes), device=dev)
    import numpy as np
    return torch.from_numpy(np.ctypeslib.as_array((C.c_ubyte * nbytes).from_address(ptr)))


def broadcast_weights(ctx, device_index: int = 0, src: int = 0, force: bool = False) -> int:
    """Broadcast the folded weights of ``ctx`` from rank ``src`` (RCCL over xGMI under backend "nccl"): every
    chunk of every weight region (``ctx.weights_regions()``) is broadcast in place, then ``ctx.weights_adopt()``
    re-reads the value-dependent layer flags that travel in the region headers.

    Every rank must have loaded the same model configurations -- non-source ranks with placeholder (zero)
    tensors.  The layouts are compared first (shape-only hash + chunk sizes, all ranks gather all ranks'
    signatures), so a mismatch raises on EVERY rank instead of leaving rank ``src`` blocked in the collective.
    Returns the number of bytes broadcast.  ``force``: run the collectives in a 1-rank group too (the single-GPU
    test of the RCCL path: device views of the library's chunks, all_gather, broadcast, adopt)."""
    regions, layout = ctx.weights_regions()
    nbytes = sum(n for _, n in regions)
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return nbytes
    on_gpu = dist.get_backend() == "nccl"
    if not on_gpu and getattr(ctx, "regions_on_device", False):
        # weights_regions() of a real context hands out hipMalloc'd addresses: wrapping them as host memory for a CPU
        # backend would read and write HBM addresses from the CPU
        raise RuntimeError(f"broadcast_weights: backend {dist.get_backend()!r} cannot move device-resident weight "
                           "regions; use backend 'nccl' (RCCL) for rvcx contexts")
    dev = torch.device("cuda", device_index) if on_gpu else torch.device("cpu")
    sizes_hash = 1469598103934665603
    for _, n in regions:
        sizes_hash = ((sizes_hash ^ n) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    sig = torch.tensor([layout & 0xFFFFFFFF, layout >> 32, sizes_hash & 0xFFFFFFFF, sizes_hash >> 32, len(regions),
                        nbytes], dtype=torch.int64, device=dev)
    sigs = [torch.empty_like(sig) for _ in range(dist.get_world_size())]
    dist.all_gather(sigs, sig)
    mine = sig.cpu().tolist()
    bad = [r for r, t in enumerate(sigs) if t.cpu().tolist() != sigs[src].cpu().tolist()]
    if bad:
        raise RuntimeError(f"weight layouts differ across ranks (ranks {bad} disagree with rank {src}; this rank: "
                           f"{len(regions)} chunks, {nbytes} bytes, hash {layout:#x}): load the same model "
                           f"configurations on every rank before broadcasting ({mine})")
    for ptr, n in regions:
        if n == 0:
            continue
        t = _view(ptr, n, dev)
        dist.broadcast(t, src)
    if on_gpu:
        torch.cuda.synchronize(dev)
    ctx.weights_adopt()
    return nbytes


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_int64(value: int, device=None) -> List[int]:
    """every rank's 64-bit value, on every rank (e.g. a digest of a probe conversion: did the broadcast weights arrive?)"""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [int(value)]
    v = int(value) & 0xFFFFFFFFFFFFFFFF
    t = torch.tensor([v - (1 << 64) if v >= (1 << 63) else v], dtype=torch.int64, device=device if device is not None else "cpu")
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [int(o.item()) & 0xFFFFFFFFFFFFFFFF for o in out]


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
