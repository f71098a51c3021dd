"""Multi-GPU sharding of a batch of independent streams: one process per GPU, rank r owns a
contiguous range of stream indices; there is NO data-path collective.  torch.distributed is used only to
bracket timing (barrier) and to reduce scalars (max elapsed time, total counts)."""
from __future__ import annotations


def shard_range(rank: int, world: int, n_total: int) -> tuple[int, int]:
    """[first, last) stream indices of `rank` when n_total streams are split as evenly as possible."""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


def max_over_ranks(value: float, device=None) -> float:
    """MAX all-reduce of a python float (works on the gloo backend with CPU tensors and on nccl/RCCL)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: int, device=None) -> int:
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return int(value)
    t = torch.tensor([value], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gather_ints(value: int, device=None) -> list[int]:
    """One integer per rank, on every rank (a SUM all-reduce of a one-hot vector: works on every backend)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [int(value)]
    t = torch.zeros(dist.get_world_size(), dtype=torch.int64, device=device if device is not None else "cpu")
    t[dist.get_rank()] = int(value)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]
