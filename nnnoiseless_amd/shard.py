"""Stream sharding across ranks: streams are independent, so each rank owns a
contiguous block and the only collective is the throughput aggregation (sum of frames, max of time)."""


def shard_range(n_streams, rank, world):
    """Contiguous, balanced block [lo, hi) of `n_streams` for `rank` of `world`."""
    base, rem = divmod(n_streams, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def aggregate(dist, frames_done, elapsed_s, device=None):
    """(total frames over all ranks, max elapsed over ranks).  `dist` = torch.distributed or None."""
    if dist is None or not dist.is_initialized():
        return frames_done, elapsed_s
    # (an initialised group of one rank still runs the two all-reduces: the 1-rank RCCL test exercises exactly this path)
    import torch
    f = torch.tensor([float(frames_done)], dtype=torch.float64, device=device)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(f.item()), float(t.item())


def gather(dist, value, device=None):
    """[value of rank 0, value of rank 1, ...] on every rank (a list of one without a process group): per-rank rates beside the
    aggregate, so that a slow GPU of a node shows in the bench line instead of hiding behind the max."""
    if dist is None or not dist.is_initialized():
        return [float(value)]
    import torch
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]
