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


_hip = None


def _hip_memcpy_d2d(dst: int, src: int, n: int):
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
    rc = _hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(n), 3)
    if rc != 0:
        raise RuntimeError(f"hipMemcpy D2D failed: {rc}")


def broadcast_weights(ctx, device_index: int, src: int = 0) -> int:
    """RCCL-broadcast the contiguous folded-weight slab of ``ctx`` from rank ``src``.  Every rank must
    have loaded models with identical layouts (non-source ranks may load zero-filled tensors).
    Returns the number of bytes broadcast."""
    ptr, nbytes = ctx.weights_blob()
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return nbytes
    dev = torch.device("cuda", device_index)
    sizes = torch.tensor([nbytes], dtype=torch.int64, device=dev)
    dist.all_reduce(sizes, op=dist.ReduceOp.MAX)
    if int(sizes.item()) != nbytes:
        raise RuntimeError("weight slab layouts differ across ranks")
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    if dist.get_rank() == src:
        _hip_memcpy_d2d(buf.data_ptr(), ptr, nbytes)
    torch.cuda.synchronize(dev)
    dist.broadcast(buf, src)
    torch.cuda.synchronize(dev)
    if dist.get_rank() != src:
        _hip_memcpy_d2d(ptr, buf.data_ptr(), nbytes)
    return nbytes


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
