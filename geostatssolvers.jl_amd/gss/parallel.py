"""Sharding of independent work over one-process-per-GPU ranks.

Domain points (kriging) and realisations (FFTGS, LUGS) are independent given the factor /
spectrum, so each rank takes a contiguous block and no data-path collective is needed.  The only
collective is an optional broadcast of the factor state from rank 0 (RCCL over xGMI when the
process group backend is `nccl`; `gloo` in the CPU tests)."""
from __future__ import annotations

from typing import Tuple


def world() -> Tuple[int, int]:
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


def shard_range(total: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `total` items for `rank`; sizes differ by at most one."""
    base, rem = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def broadcast_(tensor, src: int = 0):
    """In-place broadcast of a state tensor from `src` (no-op for a single rank)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(tensor, src=src)
    return tensor


def all_gather_concat(local, total: int):
    """Gather contiguous shards back into the full array on every rank (numpy in, numpy out)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        return local
    t = torch.as_tensor(np.ascontiguousarray(local))
    sizes = [shard_range(total, r, ws) for r in range(ws)]
    maxlen = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxlen,) + tuple(t.shape[1:]), dtype=t.dtype)
    pad[: t.shape[0]] = t
    bufs = [torch.zeros_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad)
    return np.concatenate([b[: hi - lo].numpy() for b, (lo, hi) in zip(bufs, sizes)], axis=0)
