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


# ---- launcher helpers (bench.py --gpus N outside torchrun) ---------------------------------------------------------------------
def visible_gpu_count(sysfs_root: str = "/sys/class/kfd/kfd/topology/nodes", environ=None) -> int:
    """GPUs this process would see, WITHOUT touching the HIP runtime (a launcher parent must never initialise the GPU:
    its children are the ranks).  Counts the KFD topology nodes that have SIMDs (CPU nodes have ``simd_count 0``) and
    applies ``ROCR_VISIBLE_DEVICES`` / ``HIP_VISIBLE_DEVICES`` / ``CUDA_VISIBLE_DEVICES`` the way the runtime does (a list
    of indices or UUIDs; an index beyond the device count ends the list).  Returns 0 when the topology is not readable."""
    import os
    from pathlib import Path

    env = os.environ if environ is None else environ
    n = 0
    try:
        nodes = sorted(Path(sysfs_root).iterdir(), key=lambda p: int(p.name) if p.name.isdigit() else 1 << 30)
    except OSError:
        return 0
    for node in nodes:
        try:
            props = dict(line.split(None, 1) for line in (node / "properties").read_text().splitlines() if " " in line)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):  # each filters what the previous left
        v = env.get(var)
        if v is None:
            continue
        keep = 0
        for tok in (x.strip() for x in v.split(",")):
            if tok == "":
                break
            if tok.lstrip("-").isdigit():
                if not 0 <= int(tok) < n:
                    break  # (the runtime stops at the first invalid index)
            keep += 1  # an index in range, or a UUID (GPU-xxxx): taken at face value
        n = min(n, keep)
    return n


def offset_phase_exchange_probe(partial: torch.Tensor, finish, repeats: int = 20) -> dict:
    """Time the ONE data-path collective of the engine on the live process group and check its contract: ``partial`` is this
    rank's offset-phase sums (3K + 2 floats, ``Engine.m_partial``), ``finish(reduced) -> offsets`` the closed form
    (``Engine.m_finish``).  Every rank must end up with bitwise the same offsets (all-gather + fixed-order sum).  Returns
    {us_per_exchange, backend, world_size, n_floats, offsets_bitwise_equal_across_ranks}."""
    import time

    if not is_dist():
        return {"us_per_exchange": None, "backend": None, "world_size": 1, "n_floats": int(partial.numel()),
                "offsets_bitwise_equal_across_ranks": True}
    sync = (lambda: torch.cuda.synchronize(partial.device)) if partial.is_cuda else (lambda: None)
    reduced = all_reduce_partial(partial)  # warm-up (connection set-up)
    sync()
    tdist.barrier()
    t0 = time.perf_counter()
    for _ in range(repeats):
        reduced = all_reduce_partial(partial)
    sync()
    dt = time.perf_counter() - t0
    off = finish(reduced).contiguous()
    bits = off.view(torch.int32) if off.dtype == torch.float32 else off
    parts = [torch.empty_like(bits) for _ in range(tdist.get_world_size())]
    tdist.all_gather(parts, bits)
    same = all(bool(torch.equal(parts[0], p)) for p in parts[1:])
    return {"us_per_exchange": 1e6 * dt / repeats, "backend": tdist.get_backend(), "world_size": tdist.get_world_size(),
            "n_floats": int(partial.numel()), "offsets_bitwise_equal_across_ranks": same}
