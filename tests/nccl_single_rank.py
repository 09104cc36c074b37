"""Helper of tests/test_gpu_dist.py, run as its own process on a GPU box: the RCCL leg of polgen-rvc_amd/dist.py in
a 1-rank "nccl" group -- torch tensors that alias the library's weight chunks (no copy, both directions),
all_gather of the layout signatures, in-place broadcast of every chunk, adopt, max-over-ranks and barrier."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

import polgen_rvc_amd  # noqa: F401
from polgen_rvc_amd import _lib, dist as D, synthetic as S, weights as W

os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group(backend="nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
hcfg, rcfg, scfg = S.HUBERT_CFG_TINY, S.RMVPE_CFG_TINY, S.SYNTH_CFG_TINY
ctx.load_hubert(W.hubert_cfg_struct(hcfg), S.hubert_state(hcfg, 3))
ctx.load_rmvpe(W.rmvpe_cfg_struct(rcfg), S.rmvpe_state(rcfg, 3))
sd = S.fcpe_state(S.FCPE_CFG_TINY, 401)
ctx.load_fcpe(W.fcpe_cfg_struct(W.fcpe_cfg_from_state(sd)), sd)
mid = ctx.load_synth(W.synth_cfg_struct(scfg, hcfg["embed_dim"]), S.synth_state(scfg, 3, input_dim=hcfg["embed_dim"]))
p = _lib.Params(0.0, 50.0, 1100.0, 0.0, 0.33, 1.0, 0, 1, 6, 38, 41, 7)
clip = S.make_clip(3, 2.0)
before = ctx.convert_batch(mid, [clip], p)[0]
regions, layout = ctx.weights_regions()
assert len(regions) >= 4 and all(n > 0 for _, n in regions)
# the torch view aliases the chunk: what torch writes the library reads, and back
ptr, n = regions[-1]                        # the voice model's (last) chunk
t = D._view(ptr, n, dev)
assert t.is_cuda and t.dtype == torch.uint8 and t.numel() == n and t.data_ptr() == ptr
keep = t.clone()
t[4096:].zero_()                            # everything behind the flag header
torch.cuda.synchronize()
zeroed = ctx.convert_batch(mid, [clip], p)[0]
assert not np.array_equal(zeroed, before)
t.copy_(keep)
torch.cuda.synchronize()
assert np.array_equal(ctx.convert_batch(mid, [clip], p)[0], before)
# the collectives themselves, in place on the library's memory
nbytes = D.broadcast_weights(ctx, 0, 0, force=True)
assert nbytes == sum(n for _, n in regions)
assert np.array_equal(ctx.convert_batch(mid, [clip], p)[0], before)
assert D.max_over_ranks(1.5, dev) == 1.5
D.barrier()
dist.destroy_process_group()
ctx.close()
print("NCCL_SINGLE_RANK_OK", nbytes)
