"""Sharding of independent work over one-process-per-GPU ranks.

Domain points (kriging) and realisations (FFTGS, LUGS, SGS) are independent given the factor / spectrum, so each rank
takes a contiguous block and the data path has no collective.  The reference makes the same cut: `preprocess` runs
once and `solvesingle` is mapped over realisations (/root/reference/src/simulation/fft.jl:62,145, lu.jl:76,171).
The one collective is the broadcast of the preprocess state from rank 0 (`replicate_state`): RCCL over xGMI when the
process group backend is `nccl`, `gloo` in the CPU tests; `share="native"` does the same broadcast inside libgss_hip.so
(gss_comm_init + gss_state_bcast: RCCL without torch, the route a Julia host takes).  `all_gather_concat`
(`solve(..., gather=True)`, the default: the reference's `solve` returns the whole solution) reassembles the full
result on every rank; `gather=False` leaves every rank with its block.

Collectives run on whichever device the process group serves: under an `nccl`-only group host arrays are staged
through HBM, under `gloo` device tensors are staged through host memory."""
from __future__ import annotations

from typing import Callable, Tuple


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


def _backend_map() -> dict:
    """device type -> backend name of the default process group ("cpu:gloo,cuda:nccl" -> {cpu: gloo, cuda: nccl})."""
    import torch.distributed as dist
    cfg = str(dist.get_backend_config())
    if ":" not in cfg:
        return {"cpu": cfg, "cuda": cfg} if cfg == "gloo" else {"cuda": cfg}
    return dict(part.split(":", 1) for part in cfg.split(","))


def _collective_device(t):
    """Device on which a collective over tensor `t` has to run for the default group."""
    import torch
    bm = _backend_map()
    if t.is_cuda:
        # gloo's device-tensor path is not relied upon on ROCm: stage through the host
        return t.device if bm.get("cuda") == "nccl" else torch.device("cpu")
    if "cpu" in bm:
        return t.device
    return torch.device("cuda", torch.cuda.current_device())


def broadcast_(tensor, src: int = 0):
    """In-place broadcast of a state tensor from `src` (no-op for a single rank)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensor
    dev = _collective_device(tensor)
    if dev == tensor.device:
        dist.broadcast(tensor, src=src)
    else:
        staged = tensor.to(dev)
        dist.broadcast(staged, src=src)
        if dist.get_rank() != src:
            tensor.copy_(staged)
    return tensor


def replicate_state(make: Callable[[bool], object], share: str = "broadcast"):
    """Preprocess once, realise everywhere (fft.jl:62,145; lu.jl:76,171).

    `make(compute)` builds a handle; with `compute=False` it only allocates the state.  Rank 0 computes, the peers
    receive `handle.state_tensor()` by broadcast and call `handle.adopt_state()`.  `share="recompute"` (or a single
    rank, or an engine whose handles cannot alias their state) makes every rank compute instead.  A failure on rank 0
    (e.g. a covariance that is not positive definite) is raised on every rank, not left as a hung broadcast."""
    import torch
    rank, ws = world()
    if ws == 1 or share == "recompute":
        return make(True)
    if share not in ("broadcast", "native"):
        raise ValueError(f"share={share!r}: 'broadcast', 'native' or 'recompute'")
    flag = torch.zeros(1, dtype=torch.int32)
    handle, err = None, None
    if rank == 0:
        try:
            handle = make(True)
            if not hasattr(handle, "state_tensor"):
                flag[0] = 2                     # engine without aliasable state: everybody computes
        except Exception as e:                  # noqa: BLE001 - re-raised below, after the peers have been told
            err = e
            flag[0] = 1
    broadcast_(flag, 0)
    code = int(flag.item())
    if code == 1:
        if err is not None:
            raise err
        raise RuntimeError("preprocess failed on rank 0")
    if code == 2:
        return handle if rank == 0 else make(True)
    # the peers allocate; a peer that cannot (out of memory, a different grid) must not leave rank 0 alone in the
    # broadcast: a second flag is exchanged first (max over ranks) and the failure is raised everywhere
    perr = None
    if rank != 0:
        try:
            handle = make(False)
        except Exception as e:                  # noqa: BLE001
            perr = e
    if _any_rank(perr is not None):
        raise perr if perr is not None else RuntimeError("a peer rank could not allocate the state")
    if share == "native" and hasattr(handle, "bcast_state"):
        native_comm()
        handle.bcast_state(0)                   # ncclBroadcast inside the library; the peers adopt there
        return handle
    broadcast_(handle.state_tensor(), 0)
    if rank != 0:
        try:
            handle.adopt_state()
        except Exception as e:                  # noqa: BLE001
            perr = e
    if _any_rank(perr is not None):
        raise perr if perr is not None else RuntimeError("a peer rank could not adopt the state")
    return handle


def _any_rank(flag: bool) -> bool:
    """True on every rank if `flag` is true on any (one small all-reduce on the group's device)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if flag else 0], dtype=torch.int32)
    dev = _collective_device(t)
    t = t.to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


_native_ready = False


def native_comm():
    """The library's own RCCL communicator over the ranks of the default process group (gss.h: gss_comm_init): rank 0
    draws the unique id, torch.distributed only carries its 128 bytes.  Idempotent."""
    global _native_ready
    if _native_ready:
        return
    import torch.distributed as dist
    from . import _lib
    rank, ws = world()
    box = [_lib.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    _lib.comm_init(box[0], rank, ws)
    _native_ready = True


def all_gather_concat(local, total: int):
    """Gather contiguous shards (axis 0) back into the full array on every rank.  numpy in -> numpy out, torch
    tensor in -> torch tensor on the same device out."""
    import numpy as np
    import torch
    import torch.distributed as dist
    rank, ws = world()
    if ws == 1:
        return local
    is_np = not isinstance(local, torch.Tensor)
    t = torch.as_tensor(np.ascontiguousarray(local)) if is_np else local.contiguous()
    dev = _collective_device(t)
    home = t.device
    t = t.to(dev)
    sizes = [shard_range(total, r, ws) for r in range(ws)]
    maxlen = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxlen,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(bufs, pad)
    full = torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)
    return full.cpu().numpy() if is_np else full.to(home)
