#!/usr/bin/env python3
"""Serving throughput with K rvcx contexts on ONE GPU (one host thread each, ctypes releases the GIL): the
latency-bound parts of one conversion (BiGRU, small U-Net / TextEncoder kernels) overlap the throughput-bound NSF
decoder of another.  Not the bench.py metric (that is one clip at a time); reported in DESIGN.md as the batch figure.
usage: bench_two_contexts.py [contexts=2] [clips_per_context=12]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import polgen_rvc_amd  # noqa
from polgen_rvc_amd import _lib, synthetic as S
import bench as B

K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
ctxs, mids, wavs, outs = [], [], [], []
params = B.make_params(0)
for k in range(K):
    c = _lib.Context(0)
    mids.append(B.load_models(c))
    ctxs.append(c)
    clip = S.make_clip(1000 + k, B.CLIP_SECONDS)
    wavs.append(torch.from_numpy(clip).to(dev))
    outs.append(torch.empty(c.out_capacity(mids[k], clip.shape[0], params), dtype=torch.int16, device=dev))
n = wavs[0].shape[0]


def worker(k, reps):
    for _ in range(reps):
        ctxs[k].convert_batch_raw(mids[k], [wavs[k].data_ptr()], [n], params, [outs[k].data_ptr()])


for k in range(K):
    worker(k, 2)                      # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
th = [threading.Thread(target=worker, args=(k, N)) for k in range(K)]
for t in th:
    t.start()
for t in th:
    t.join()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{K} contexts x {N} clips of {B.CLIP_SECONDS:.0f} s: {dt*1e3/(K*N):.2f} ms per clip, aggregate RTF {K*N*B.CLIP_SECONDS/dt:.0f}x")
