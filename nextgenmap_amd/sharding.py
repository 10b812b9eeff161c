"""Read sharding + the one collective of the path.

Reads (pairs of reads for paired-end input) are independent units: rank r of W owns one contiguous index
range, the genome/index are replicated per GPU, nothing is exchanged per batch.  At end of run the
mapping statistics are summed with a single all-reduce (RCCL over xGMI on GPUs, gloo in the CPU tests).
NextGenMap itself only has per-thread, last-writer-wins counters (src/NGM.cpp:172-200,
src/AlignmentBuffer.cpp:209-212); true sums are reported here.
"""
STAT_NAMES = ("reads", "mapped", "unmapped", "written", "pairs_total", "pairs_broken", "insert_sum", "insert_cnt")


def shard_range(n_reads, rank, world, paired=False):
    """Contiguous [lo, hi) owned by `rank`; with paired=True boundaries fall on even indices so mates stay together."""
    unit = 2 if paired else 1
    n_units = (n_reads + unit - 1) // unit
    base, extra = divmod(n_units, world)
    lo_u = rank * base + min(rank, extra)
    hi_u = lo_u + base + (1 if rank < extra else 0)
    return min(lo_u * unit, n_reads), min(hi_u * unit, n_reads)


def reduce_stats(local, device=None):
    """Sum the int64[8] stats vector over all ranks with ONE all-reduce; returns a dict on every rank."""
    import torch
    import torch.distributed as dist
    v = torch.tensor([int(local.get(k, 0)) for k in STAT_NAMES], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(v, op=dist.ReduceOp.SUM)
    return dict(zip(STAT_NAMES, (int(x) for x in v.tolist())))
