"""The RCCL leg of the multi-GPU path on ONE GPU (no 8-GPU node in this pool): a fresh process forms a 1-rank "nccl"
group and runs polgen-rvc_amd/dist.py's weight broadcast on the library's own device chunks (tests/nccl_single_rank.py).
What this does not cover -- more than one rank -- is covered on CPU by tests/test_dist_gloo.py (world sizes 2 and 4)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_rccl_broadcast_path_in_a_single_rank_group():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "nccl_single_rank.py"), str(port)], capture_output=True,
                       text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-4000:])
    assert r.returncode == 0 and "NCCL_SINGLE_RANK_OK" in r.stdout


def test_two_rank_rccl_broadcast_through_bench_py():
    """The first box with more than one GPU exercises the real thing: `bench.py --gpus 2` starts one
    worker per GPU, rank 1 loads shape-only placeholders, receives rank 0's weight regions and the index by RCCL broadcast,
    and both ranks convert the same probe clip: equal PCM digests = the broadcast weights are the ones rank 0 folded.
    Skipped on the 1-GPU boxes of this pool (torch.cuda.device_count() does not initialise the GPU on this image)."""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "c3", "--batch", "4",
                        "--steps", "1", "--warmup", "1", "--no-roofline", "--verify-ranks"], capture_output=True, text=True,
                       timeout=1500, cwd=root)
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["weights_bcast_bytes"] > 0
    assert d["config"]["ranks_agree"] is True and len(d["config"]["rank_pcm_digests"]) == 2
    assert "C4" in d["config"]["workload"] and d["value"] > 0
