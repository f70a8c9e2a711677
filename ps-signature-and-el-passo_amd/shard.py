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


def rank_memory_budget(curve, attrs, hidden, window_bits, batch, hbm_bytes=288 * 10**9):
    """Device memory one rank needs for a verification workload (what `bench.py --dry-run` prints per rank, and what the real run's `key_tables.table_bytes`
    must equal): the key's signed-digit fixed-base tables (csrc/elp/curve.h: ceil(256 / W) windows x 2^(W-1) entries per base; A + 6 G1 and A + 2 G2 bases,
    affine Montgomery limbs), the per-lane launch workspace of the tables of small multiples (csrc/elp/pipeline.h vtab_words: 8 G2 + 24 G1 entries padded to
    16 bytes) and the batch (records in, one verdict byte out).  Everything is per GPU: shards share nothing."""
    nl, fbytes = (9, 32) if curve in ("bn254", 0) else (14, 48)
    w = window_bits or 8
    nwin, per = -(-256 // w), 1 << (w - 1)
    g1_entry, g2_entry = 2 * nl * 4, 4 * nl * 4
    tables = nwin * per * ((attrs + 6) * g1_entry + (attrs + 2) * g2_entry)
    pad16 = lambda x: -(-x // 16) * 16      # noqa: E731
    workspace = batch * (8 * pad16(g2_entry) + 24 * pad16(g1_entry))
    rec = 5 * 2 * fbytes + 4 * fbytes + 32 * (1 + hidden + 2 + (attrs - hidden))
    total = tables + workspace + batch * (rec + 1)
    return {"tables_bytes": tables, "workspace_bytes": workspace, "records_bytes": batch * rec, "total_bytes": total, "hbm_bytes": hbm_bytes,
            "fraction_of_hbm": total / hbm_bytes, "fits": total < 0.9 * hbm_bytes}
