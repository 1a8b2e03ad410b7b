"""Sample sharding across GPUs -- the reference's only parallelism is a multiprocessing.Pool over
samples (tredparse/tred.py:528-532); here it is one process per GPU with the same partitioning
unit (a sample with all its loci stays on one device, so its BAM is opened once).

No data-path collective exists: sample x locus units are independent.  torch.distributed (RCCL on the
GPUs, gloo in the CPU tests) carries only the barrier and two scalar reductions of the report.
"""


def shard_range(n_samples, rank, world):
    """Contiguous block partition of sample indices: [start, end) of this rank."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_samples, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def sample_owner(sample_index, n_samples, world):
    """Inverse of shard_range."""
    base, extra = divmod(n_samples, world)
    cut = extra * (base + 1)
    if sample_index < cut:
        return sample_index // (base + 1)
    return extra + (sample_index - cut) // base


def aggregate(units_local, elapsed_local, dist=None, device=None):
    """Whole-job throughput: (sum of units over ranks) / (max of elapsed over ranks)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return units_local, elapsed_local
    import torch
    t = torch.tensor([elapsed_local], dtype=torch.float64, device=device)
    u = torch.tensor([units_local], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return int(u.item()), float(t.item())
