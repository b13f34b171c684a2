"""Frame sharding for multi-GPU runs: frames are independent, so a batch is split into
contiguous per-rank blocks and there is no collective on the data path.  The only collectives are the
benchmark's barrier and the MAX-reduce of the elapsed time."""
from __future__ import annotations


def shard_range(n_frames: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank `rank`; sizes differ by at most one, earlier ranks larger."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def frame_seeds(base_seed: int, rank: int, world: int, frames_per_rank: int):
    """Weak-scaling workload: every rank gets `frames_per_rank` frames, frame k of the global batch uses
    seed base_seed + k ('frame k uses seed 12345 + k')."""
    lo, hi = shard_range(frames_per_rank * world, rank, world)
    return [base_seed + k for k in range(lo, hi)]


def max_over_ranks(value: float, device=None) -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
