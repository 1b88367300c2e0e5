"""Batch sharding over the GPUs of one node: one process per GPU, clips split evenly, weights replicated.

The forward is embarrassingly parallel over clips (eval-mode BatchNorm mixes no samples), so the
data path needs no collective.  The only exchange is one all-gather of the per-clip loss (``B/R`` floats per
rank; RCCL on ROCm via ``torch.distributed`` backend "nccl", gloo in the CPU tests).
"""
from __future__ import annotations

import inspect
import os

import torch
import torch.distributed as dist


def shard_range(total: int, rank: int, world: int):
    """Half-open clip range of ``rank``: ``[rank*total/world, (rank+1)*total/world)``; requires divisibility."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank/world {rank}/{world}")
    if total % world:
        raise ValueError(f"batch {total} is not divisible by world size {world}")
    per = total // world
    return rank * per, (rank + 1) * per


def init_from_env(backend: str | None = None):
    """Initialise ``torch.distributed`` from torchrun's environment; returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kwargs = {}
        if backend == "nccl":
            # one process per GPU: bind this rank to its device BEFORE the communicator exists, so that barriers and
            # collectives never guess a device (RCCL would otherwise start every rank on device 0)
            torch.cuda.set_device(local_rank)
            # decided from the signature, not by catching TypeError: a TypeError raised inside a half-finished
            # initialisation must surface as itself, not as the retry's "already initialized"
            if "device_id" in inspect.signature(dist.init_process_group).parameters:
                kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def rank() -> int:
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size() -> int:
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def gather_per_clip(local: torch.Tensor) -> torch.Tensor:
    """All-gather equally sized per-clip vectors in rank order; identity when not distributed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out


def max_over_ranks(value: float, device) -> float:
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
