"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL on ROCm).

The q_phase shards by clips (independent warm-start chains): contiguous blocks of
clips per rank, no data-path collective; results are gathered to rank 0 for packaging (or stay as
per-rank shards; a 1 M-frame run is 2.7 GB of outputs, which no other rank needs).  The
offset phase has one real exchange step: the per-rank partial sums ``[s(3K), z2, T]`` are combined
with ONE all-reduce of 3K+2 floats (``stac_core.py:157-160`` summed over ranks).  For run-to-run
determinism ``deterministic=True`` uses all-gather + fixed-order sum instead of a ring reduce.

Works with any initialised process group (tests use gloo / world_size 2 on CPU).
"""

from __future__ import annotations

import torch

try:
    import torch.distributed as tdist
except Exception:  # pragma: no cover
    tdist = None


def is_dist() -> bool:
    return tdist is not None and tdist.is_available() and tdist.is_initialized()


def world() -> tuple[int, int]:
    """(rank, world_size); (0, 1) when no process group is initialised."""
    if is_dist():
        return tdist.get_rank(), tdist.get_world_size()
    return 0, 1


def barrier() -> None:
    """No-op without a process group."""
    if is_dist():
        tdist.barrier()


def shard_range(n_items: int, rank: int | None = None, world_size: int | None = None) -> tuple[int, int]:
    """Contiguous block [lo, hi) of ``n_items`` clips owned by ``rank`` (earlier ranks get the remainder)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_reduce_partial(partial: torch.Tensor, deterministic: bool = True) -> torch.Tensor:
    """Sum the offset-phase partial sums over ranks (the only data-path collective of the engine)."""
    if not is_dist():
        return partial
    if deterministic:
        parts = [torch.empty_like(partial) for _ in range(tdist.get_world_size())]
        tdist.all_gather(parts, partial.contiguous())
        out = parts[0].clone()
        for p in parts[1:]:  # fixed rank order -> bitwise reproducible
            out = out + p
        return out
    out = partial.clone()
    tdist.all_reduce(out, op=tdist.ReduceOp.SUM)
    return out


def all_gather_clips(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """Concatenate per-rank clip blocks (dim 0) in rank order on every rank; handles ragged shards."""
    if not is_dist():
        return local
    w = tdist.get_world_size()
    sizes = [shard_range(n_total, r, w) for r in range(w)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(w)]
    tdist.all_gather(bufs, pad)
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)


def gather_clips(local: torch.Tensor, n_total: int, dst: int = 0):
    """Per-rank clip blocks (dim 0) concatenated in rank order on rank ``dst`` only; other ranks get None.
    Ragged shards are padded to the largest one for the collective."""
    if not is_dist():
        return local
    w, r = tdist.get_world_size(), tdist.get_rank()
    sizes = [shard_range(n_total, k, w) for k in range(w)]
    maxn = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((maxn,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(w)] if r == dst else None
    tdist.gather(pad, bufs, dst=dst)
    if r != dst:
        return None
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=0)
