"""Process-group bootstrap: one process per GPU, RCCL over xGMI.

``init_distributed()`` reads the torchrun environment (``RANK``, ``WORLD_SIZE``,
``LOCAL_RANK``, ``MASTER_ADDR``/``MASTER_PORT``), binds the process to its GPU and
creates the default group with backend ``nccl`` (= RCCL on ROCm) on a GPU box or
``gloo`` on a CPU-only box.  With gloo, collectives only ever see CPU tensors
(``to_host``/``from_host`` helpers copy synchronously).
"""
from __future__ import annotations

import dataclasses
import os
from typing import Optional

import torch
import torch.distributed as dist


@dataclasses.dataclass
class DistInfo:
    rank: int = 0
    world: int = 1
    local_rank: int = 0
    backend: str = "none"
    device: torch.device = torch.device("cpu")

    @property
    def is_main(self) -> bool:
        return self.rank == 0


def init_distributed(backend: Optional[str] = None, use_gpu: Optional[bool] = None) -> DistInfo:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    gpu = torch.cuda.is_available() if use_gpu is None else use_gpu
    device = torch.device("cuda", local) if gpu else torch.device("cpu")
    if gpu:
        torch.cuda.set_device(device)
    if world == 1:
        return DistInfo(rank, world, local, "none", device)
    backend = backend or ("nccl" if gpu else "gloo")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if not dist.is_initialized():
        kw = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return DistInfo(rank, world, local, backend, device)


def shutdown() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def all_reduce_scalar(value: float, op=None, group=None, device=None) -> float:
    """Reduce one python float over the group (identity when not distributed)."""
    if not (dist.is_available() and dist.is_initialized()):
        return value
    backend = dist.get_backend(group)
    t = torch.tensor([value], dtype=torch.float64,
                     device=device if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=op or dist.ReduceOp.SUM, group=group)
    return float(t.item())


def broadcast_object(obj, src: int = 0, group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]
