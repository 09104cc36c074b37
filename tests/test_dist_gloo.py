"""CPU, world_size 2 over gloo: the N>1 plumbing of bench.py / polgen-rvc_amd/dist.py -- utterance
sharding without collectives in the data path, broadcast of a weight blob from rank 0, max-over-ranks
timing."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import polgen_rvc_amd  # noqa: F401
    from polgen_rvc_amd import dist as D
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, l, w = D.init("gloo")
    assert (r, w) == (rank, world)
    blob = torch.arange(1000, dtype=torch.float32) if rank == 0 else torch.zeros(1000)
    D.broadcast_tensor(blob, 0)
    mine = D.shard(7, rank, world, [3, 9, 1, 9, 4, 4, 2])
    t = D.max_over_ranks(1.0 + rank)
    D.barrier()
    q.put((rank, float(blob.sum()), mine, t))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == float(sum(range(1000)))
    assert sorted(res[0][2] + res[1][2]) == list(range(7))
    assert res[0][3] == res[1][3] == 2.0
