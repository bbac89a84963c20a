"""The N > 1 path of bench.py on CPU: stream sharding + the throughput aggregation collective, world_size 2, gloo."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nnnoiseless_amd.shard import aggregate, shard_range


def test_shard_range_partitions():
    for n in (1, 7, 64, 4096, 262144):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(100, rank, world)
    frames, t = aggregate(dist, (hi - lo) * 10, 1.0 + rank)
    q.put((rank, frames, t))
    dist.barrier()
    dist.destroy_process_group()


def test_aggregate_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res == [(0, 1000.0, 2.0), (1, 1000.0, 2.0)]  # sum of frames, max of elapsed
