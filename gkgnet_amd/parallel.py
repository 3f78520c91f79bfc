"""Data-parallel plumbing: one process per GPU, gradients averaged with ONE flat all-reduce.

The path shards over the batch only (every (b, g) k-NN problem is independent — reference
torch_vertex.py:199-202); the single exchange step is the gradient all-reduce the reference gets from
MMDistributedDataParallel (mmcls/apis/train.py:117-125).  On ROCm ``backend='nccl'`` is RCCL over xGMI.
All parameter gradients live in one contiguous bucket (``p.grad`` are views into it), so the step issues a
single large collective instead of one per tensor — the right shape for point-to-point xGMI links.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Initialises the default process group from the torchrun environment.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("GKG_DIST_BACKEND") == "gloo" and torch.cuda.is_available():
        local %= torch.cuda.device_count()             # ranks may share a device in the single-GPU exercise
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # GKG_DIST_BACKEND=gloo lets a multi-rank run share ONE GPU (RCCL refuses duplicate devices): used to
            # exercise the N>1 control flow on a single-GPU box; production is RCCL.
            backend = os.environ.get("GKG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


class GradBucket:
    """Flat gradient storage for a set of parameters + one all-reduce(avg) per step."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        self._zeros = None
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def zero(self):
        self.flat.zero_()

    def release(self):
        """Detach ``p.grad`` from the bucket so the next backward writes fresh gradients (no zero-fill and no
        per-parameter accumulate kernels); follow the backward with :meth:`pack`."""
        for p in self.params:
            p.grad = None

    def pack(self):
        """Gather the freshly produced ``p.grad`` tensors into the flat bucket with one batched copy and
        re-point ``p.grad`` at the bucket views."""
        if self._zeros is None:          # parameters whose gradient is identically zero come back as None
            self._zeros = torch.zeros(max(p.numel() for p in self.params), device=self.flat.device, dtype=self.flat.dtype)
        grads = [(p.grad.reshape(-1) if p.grad is not None else self._zeros[:p.numel()]) for p in self.params]
        torch.cat(grads, out=self.flat)
        o = 0
        for p in self.params:
            p.grad = self.flat[o:o + p.numel()].view_as(p)
            o += p.numel()

    def all_reduce(self, async_op: bool = False):
        """Average over ranks.  No-op in a single process."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return None
        self.flat.div_(dist.get_world_size())
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)


def shard_batch(global_batch: int, rank: int, world: int) -> range:
    """Contiguous slice of a global batch owned by ``rank`` (remainder spread over the first ranks)."""
    base, rem = divmod(global_batch, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def broadcast_parameters(module: torch.nn.Module, src: int = 0):
    """Make every rank start from rank ``src``'s parameters and buffers (what DDP does at construction)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
