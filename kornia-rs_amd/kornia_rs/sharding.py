"""Batch sharding across the GPUs of one node — the only multi-GPU logic the path needs.

Frames / images are independent units: rank ``g`` of ``G`` owns the contiguous
slice ``[g*N/G, (g+1)*N/G)`` on its own device and stream; there is NO data-path collective.
``torch.distributed`` (RCCL on GPUs, gloo in the CPU tests) is used only to line ranks up at the
start of a timed region and to take the slowest rank's time.
"""
from __future__ import annotations

from typing import Tuple


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: the first ``n_items % world`` ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world: {rank}/{world}")
    if n_items < 0:
        raise ValueError("negative item count")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def aggregate_throughput(units_this_rank: float, elapsed_s: float, dist=None, device=None) -> Tuple[float, float]:
    """(total units over all ranks, slowest rank's elapsed seconds).  ``dist`` is an initialised
    ``torch.distributed`` module or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return units_this_rank, elapsed_s
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units_this_rank], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()), float(t.item())
