"""Multi-GPU host logic for the EMA-VFI path: one process per GPU, frame pairs sharded
contiguously, no collective inside the forward.

The reference has no distributed code at all (single process, single device; its ``warp`` even
hard-codes ``.cuda()``, ema_vfi.py:159-160).  What this module adds is exactly what
BASELINE.json asks for: ONE broadcast of the packed weight blob from rank 0 (RCCL over xGMI when
the backend is "nccl"; gloo in the CPU tests) and a max-over-ranks reduction of wall time for
the benchmark.  Every function works with any initialised torch.distributed backend and
degrades to a no-op for a single process.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank_world():
    """(rank, world_size, local_rank) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend: str, device=None):
    """Initialise torch.distributed from the torchrun environment; rendezvous on 127.0.0.1 unless
    MASTER_ADDR says otherwise (container hostnames may not resolve)."""
    rank, world, _ = env_rank_world()
    if world == 1 or dist.is_initialized():
        return rank, world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL on this driver
    kwargs = {"device_id": device} if (backend == "nccl" and device is not None) else {}
    dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of ``n_items`` frame pairs owned by ``rank``: sizes differ by at
    most one, earlier ranks take the remainder, every item belongs to exactly one rank."""
    if world < 1 or not (0 <= rank < world) or n_items < 0:
        raise ValueError(f"bad shard request: n_items={n_items} rank={rank} world={world}")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_packed(blob: torch.Tensor, src: int = 0) -> torch.Tensor:
    """The path's one collective: broadcast the packed weight blob (uint8) from ``src`` in place."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(blob, src=src)
    return blob


def share_model_weights(model, dtype, device):
    """Rank 0 packs its parameters; every rank ends up with the identical packed blob installed.
    Returns the blob (for checks)."""
    from . import lib
    dt = lib.dtype_code(dtype)
    rank = dist.get_rank() if dist.is_initialized() else 0
    nbytes = lib.load().emavfi_packed_bytes(model.in_channels, model.mid_channels, model.num_blocks, dt)
    blob = model.packed_weights(dt, device) if rank == 0 else torch.empty(nbytes, dtype=torch.uint8, device=device)
    broadcast_packed(blob, 0)
    if rank != 0:
        model.load_packed_weights(dt, blob)
    return blob


def max_over_ranks(value: float, device=None) -> float:
    """MAX-reduce a host scalar (the benchmark's elapsed time) over all ranks."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return float(value)
    on_cpu = device is None or dist.get_backend() == "gloo"
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if on_cpu else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_floats(values, device=None):
    """All-gather a short list of host floats over the initialised backend (device tensors over RCCL when it is "nccl"): one row
    per rank, in rank order.  bench.py's rank census - which ranks the collective backend actually saw, and each one's step time."""
    row = [float(v) for v in values]
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [row]
    on_cpu = device is None or dist.get_backend() == "gloo"
    t = torch.tensor(row, dtype=torch.float64, device="cpu" if on_cpu else device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.cpu()] for o in out]


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
