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
