"""Multi-GPU sharding of a batch of independent items: one process per GPU, contiguous shards, no
data-path collective; the only exchange is the all-reduce of the accepted-count (RCCL over xGMI on GPUs, gloo in CPU tests)."""


def shard_range(n_total, rank, world):
    """Contiguous shard [start, start + count) of rank; the first (n_total % world) ranks take one extra item."""
    base, extra = divmod(n_total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def reduce_count(local_count, dist=None, device=None):
    """Sum of the per-rank accepted counters (int64 all-reduce).  `dist` = torch.distributed if a process group is up."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(local_count)
    import torch
    t = torch.tensor([int(local_count)], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def verify_sharded(verify_fn, n_total, rank, world, dist=None, device=None):
    """Runs verify_fn(start, count) -> (flags, accepted) on this rank's shard and reduces the count.
    Returns (start, flags, local_accepted, global_accepted)."""
    start, count = shard_range(n_total, rank, world)
    flags, acc = verify_fn(start, count)
    return start, flags, int(acc), reduce_count(acc, dist, device)
