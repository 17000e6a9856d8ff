"""CPU, world_size 2 over gloo: the N>1 path of bench.py (shard + one-time weight broadcast)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kokorox_amd import dist as kd


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [kd.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert kd.shard_list(list(range(10)), 1, 3) == [4, 5, 6]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, path, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        buf = kd.broadcast_blob(path if rank == 0 else "", torch.device("cpu"), rank, world)
        want = np.fromfile(path, dtype=np.uint8)
        ok = np.array_equal(buf.numpy(), want)
        # the bench's aggregation: per-rank wall -> MAX, work -> SUM of shards
        lo, hi = kd.shard_range(129, rank, world)
        wall = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(wall, op=dist.ReduceOp.MAX)
        work = torch.tensor([float(hi - lo)], dtype=torch.float64)
        dist.all_reduce(work, op=dist.ReduceOp.SUM)
        q.put((rank, ok, float(wall.item()), float(work.item())))
    finally:
        dist.destroy_process_group()


def test_two_rank_blob_broadcast_and_aggregation(tmp_path):
    path = str(tmp_path / "blob.bin")
    np.random.default_rng(0).integers(0, 256, size=1_000_003, dtype=np.uint8).tofile(path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert all(r[2] == 2.0 and r[3] == 129.0 for r in res)
