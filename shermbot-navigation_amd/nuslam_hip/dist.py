"""Multi-GPU plumbing for the Monte-Carlo batch: one process per GPU, independent filters
sharded over ranks with NO collective in the data path; the only exchange is the end-of-run batch statistics,
gathered over RCCL/xGMI (backend "nccl" on ROCm; "gloo" in the CPU tests) and summed in rank order so the result is
bit-reproducible and independent of the ring order."""
import numpy as np


def shard(total_filters, world, rank):
    """Contiguous block partition: rank g owns filters [start, start + count)."""
    base, rem = divmod(total_filters, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def replica_seed(base_seed, global_filter_index):
    """Seed of one Monte-Carlo trial: a function of the GLOBAL filter index only, so a filter's trace (and hence its
    result, bit for bit) does not depend on how many GPUs the batch is spread over."""
    return int(base_seed) + int(global_filter_index)


def reduce_stats(local_stats, device=None):
    """all_gather the per-rank statistics vector {sum state, sum state^2, sum trace(P), count} and add the rows in
    rank order.  Returns (total, per_rank) as float64 numpy arrays; without an initialised process group returns the
    input unchanged."""
    import torch
    import torch.distributed as dist
    loc = np.ascontiguousarray(local_stats, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return loc.copy(), loc[None, :].copy()
    t = torch.from_numpy(loc.copy())
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    per_rank = np.stack([x.cpu().numpy() for x in parts])
    total = np.zeros_like(loc)
    for r in range(per_rank.shape[0]):      # fixed order: deterministic to the last bit
        total = total + per_rank[r]
    return total, per_rank


def max_over_ranks(seconds, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
