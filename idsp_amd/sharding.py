"""Lane sharding for multi-GPU runs (one process per GPU).

Lanes never interact (`Lanes::process` touches `state[i]`, `x[i]` only,
dsp-process/src/compose.rs:468-476), so N GPUs take N contiguous lane blocks
and there is no data-path collective: the only cross-rank operations are the
start/stop barrier and an optional all-reduce of an 8-byte output checksum.
"""
from __future__ import annotations

from typing import Tuple


def lane_shard(lanes: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `lanes` owned by `rank` of `world`:
    GPU g gets lanes [g*L/G, (g+1)*L/G)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("need 0 <= rank < world")
    if lanes < 0:
        raise ValueError("lanes must be non-negative")
    return lanes * rank // world, lanes * (rank + 1) // world


def checksum_i64(t) -> int:
    """Order-independent 64-bit wrapping sum of a tensor's elements viewed as
    32-bit words — summed over shards it equals the checksum of the whole."""
    import torch

    w = t.reshape(-1).view(torch.int32)
    return int(w.sum(dtype=torch.int64).item())  # accumulated in int64 (no widened copy); wraps modulo 2^64


def allreduce_checksum(value: int, device=None) -> int:
    """Sum `value` (mod 2^64) over all ranks of the default process group."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([value], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
