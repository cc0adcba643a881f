"""Scenario sharding across the GPUs of a node: contiguous shards, no per-step exchange, one final
gather of the per-scenario results (RCCL over xGMI when the backend is ``nccl``; ``gloo`` in the CPU
tests).  Scenarios are independent optimisations (ur_mpc.py:665-692 touches only its own arguments),
so the gathered array equals the single-process result row for row."""
from __future__ import annotations

from typing import Tuple

import torch


def shard_range(total: int, world: int, rank: int) -> Tuple[int, int]:
    """(first, count) of rank's contiguous shard; the first ``total % world`` ranks take one extra."""
    if not (0 <= rank < world):
        raise ValueError("rank outside the world")
    base, extra = divmod(total, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def gather_rows(local: torch.Tensor, total: int, group=None) -> torch.Tensor:
    """All ranks contribute ``local`` ([count_r, ...], shards in rank order) and receive the
    [total, ...] concatenation.  Uneven shards are padded to the largest one for the collective."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [shard_range(total, world, r)[1] for r in range(world)]
    if local.shape[0] != counts[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} rows, its shard has {counts[rank]}")
    biggest = max(counts)
    tail = tuple(local.shape[1:])
    if all(c == biggest for c in counts):
        out = torch.empty((total,) + tail, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros((biggest,) + tail, dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
