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


def _shard_worker(rank, world, port, q, lib_path):
    """One rank of the sharded data path: its block of streams through the kernels (CPU SIMT interpreter build of
    the product sources, tests/hostsim), outputs gathered to rank 0.  No data-path collective."""
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import nnnoiseless_amd as nn
    from nnnoiseless_amd import _ffi
    from nnnoiseless_amd.synthetic import make_streams
    x = make_streams(77, 7, 3)                       # every rank derives the same global input, keeps its block
    lo, hi = shard_range(7, rank, world)
    bd = nn.BatchDenoiser(hi - lo, lib=_ffi.Library(lib_path))
    out, vad = bd.process(x[lo:hi])
    got = [None] * world
    dist.all_gather_object(got, (lo, hi, out, vad))
    frames, _ = aggregate(dist, (hi - lo) * 3, 1.0)
    if rank == 0:
        q.put((got, frames))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_streams_match_unsharded_world2():
    """A stream's result must not depend on which rank owns it or where it sits in a 64-stream tile."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "hostsim"))
    import build_hostsim
    import nnnoiseless_amd as nn
    from nnnoiseless_amd import _ffi
    from nnnoiseless_amd.synthetic import make_streams
    lib_path = build_hostsim.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q, lib_path)) for r in range(2)]
    for p in procs:
        p.start()
    got, frames = q.get(timeout=600)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    x = make_streams(77, 7, 3)
    out, vad = nn.BatchDenoiser(7, lib=_ffi.Library(lib_path)).process(x)
    assert frames == 21.0
    assert [(lo, hi) for lo, hi, _, _ in got] == [(0, 4), (4, 7)]
    assert np.array_equal(np.concatenate([g[2] for g in got]), out)
    assert np.array_equal(np.concatenate([g[3] for g in got], axis=1), vad)


def _run_bench(extra, env_extra=None):
    """bench.py on CPU: gloo, CPU tensors, the SIMT-interpreter build of the product sources (plumbing only)."""
    import json
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "hostsim"))
    import build_hostsim
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NNN_LIBRARY=build_hostsim.build())
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-run"] + extra, env=env, capture_output=True, text=True,
                       timeout=900)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, [json.loads(l) for l in lines], p.stderr


def test_bench_self_spawns_two_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself, counts them with an all-reduce and
    reports the whole-job aggregate (it used to start ONE rank and print n_gpus 2)."""
    rc, lines, err = _run_bench(["--gpus", "2"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1                               # rank 0 only
    d = lines[0]
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2
    assert d["config"]["streams_total"] == 2 * d["config"]["streams_per_gpu"]
    assert d["outputs_finite"] and d["value"] > 0
    # every rank's own rate beside the aggregate (a slow GPU of a node shows): one entry per rank, none slower than the whole job's share
    assert len(d["per_rank_frames_per_s"]) == 2 and all(r > 0 for r in d["per_rank_frames_per_s"])
    assert min(d["per_rank_frames_per_s"]) >= 0.4 * d["value"]          # (>= value / 2 by construction; whole frames per second at the interpreter's pace)


def test_bench_refuses_wrong_world():
    rc, lines, err = _run_bench(["--gpus", "2"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 2 and not lines and "refusing" in err


def test_bench_pool_wrap_stays_in_bounds():
    """A step longer than the resident pool wraps as often as needed and never runs past the pool's end (VERDICT r1:
    the old wrap ran 7 frames past x / y / vad at 65 536 streams)."""
    rc, lines, err = _run_bench(["--streams", "3", "--frames-per-step", "3", "--steps", "2", "--warmup", "1",
                                 "--pool-bytes", str(3 * 480 * 4 * 2)])          # room for 2 frames: every step wraps
    assert rc == 0, err[-2000:]
    assert lines[0]["pool_frames"] == 2 and lines[0]["outputs_finite"]
