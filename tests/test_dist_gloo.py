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


class _HostCtx:
    """Stands in for a _lib.Context on CPU: weight chunks are host arrays, weights_regions() hands out their
    addresses exactly like the device pointers of rvcx_weights_regions."""

    def __init__(self, sizes, fill, layout):
        import numpy as np
        self.chunks = [np.full(n, fill, np.uint8) if fill is not None else
                       (np.arange(n, dtype=np.int64) * (i + 3) % 251).astype(np.uint8) for i, n in enumerate(sizes)]
        self.layout, self.adopted = layout, 0

    def weights_regions(self):
        return [(c.ctypes.data, c.nbytes) for c in self.chunks], self.layout

    def weights_adopt(self):
        self.adopted += 1


def _bcast_worker(rank, world, port, q, mismatch):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    import polgen_rvc_amd  # noqa: F401
    from polgen_rvc_amd import dist as D
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    D.init("gloo")
    sizes = [4096, 1 << 20, 12345, 256]
    layout = 0xDEADBEEFCAFEF00D
    if mismatch and rank == world - 1:
        sizes = [4096, 1 << 20, 12345 + 256, 256]        # a layer more on one rank
    # rank 0 holds the folded weights, the others loaded placeholders (zeros)
    ctx = _HostCtx(sizes, None if rank == 0 else 0, layout)
    try:
        n = D.broadcast_weights(ctx, rank, 0)
        want = _HostCtx(sizes, None, layout)
        ok = all(np.array_equal(a, b) for a, b in zip(ctx.chunks, want.chunks))
        q.put((rank, "ok", n, ok, ctx.adopted))
    except RuntimeError as e:
        q.put((rank, "raised", str(e)[:60], False, ctx.adopted))
    D.barrier()
    torch.distributed.destroy_process_group()


def _run_bcast(world, mismatch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_bcast_worker, args=(r, world, port, q, mismatch)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_broadcast_weights_two_and_four_ranks():
    """dist.broadcast_weights itself (the only collective of the path): ranks that start zero-filled end equal to
    rank 0, chunk by chunk, and re-read the region flags afterwards."""
    for world in (2, 4):
        res = _run_bcast(world, mismatch=False)
        assert [r[1] for r in res] == ["ok"] * world
        assert all(r[3] for r in res) and all(r[4] == 1 for r in res)
        assert len({r[2] for r in res}) == 1 and res[0][2] == 4096 + (1 << 20) + 12345 + 256


def test_broadcast_weights_layout_mismatch_fails_on_every_rank():
    """a rank whose layout differs must make ALL ranks raise before the collective (nobody blocks in
    dist.broadcast, nothing is scrambled)."""
    res = _run_bcast(2, mismatch=True)
    assert [r[1] for r in res] == ["raised", "raised"]
    assert all(r[4] == 0 for r in res)


def _bench(*args, timeout=300):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], capture_output=True, text=True,
                          timeout=timeout, env=env)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` invoked plainly (the shape of the driver's N = 1 command, VERDICT r2 missing#6): the
    parent starts one worker per rank before touching any GPU, the workers rendezvous (gloo in this CPU dry run, RCCL
    on a GPU node), shard BASELINE configs[4]'s 256 utterances by length without overlap, and the parent relays
    exactly ONE JSON line -- rank 0's."""
    import json
    r = _bench("--gpus", "2", "--dry-run", "--workload", "c5", "--verify-ranks")
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True
    assert d["rank_digests"] == [1 << 63, (1 << 63) + 7]              # --verify-ranks: every rank's 64-bit digest, unsigned
    assert d["utterances_per_rank"] == [128, 128]                       # +-0 items per rank
    a, b = d["audio_seconds_per_rank"]
    assert abs(a - b) / (a + b) < 0.02                                  # length-sorted round-robin balances audio
    assert abs(a + b - 256 * 9.0) < 256 * 1.0                           # U(3, 15) s


def test_bench_launcher_propagates_a_failing_rank():
    r = _bench("--gpus", "2", "--dry-run", "--dry-run-fail-rank", "1")
    assert r.returncode != 0
    assert "workers failed" in r.stderr


def test_bench_refuses_more_gpus_than_visible():
    r = _bench("--gpus", "3")                    # no GPU in the CPU container / one on the test box
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)
