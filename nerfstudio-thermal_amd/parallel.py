"""Data-parallel training: one process per GPU, rays sharded by rank, ONE gradient all-reduce per step.

The reference wraps the model in torch DDP (pipelines/base_pipeline.py:281-283), whose reducer all-reduces every
parameter gradient in 25 MB buckets over NCCL.  Here the gradients already live in one flat fp32 arena
(arena.py), so the exchange is a single RCCL all-reduce (mean) over the live range -- 77.6 MB in shared mode --
or, optionally, a few large chunks so that the tail of the reduction overlaps the first Adam launches.
Works with backend "nccl" (= RCCL over xGMI on ROCm) and, for CPU tests, "gloo".
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Initialise torch.distributed from the torchrun environment.  Returns (rank, local_rank, world_size)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class GradAllReducer:
    """Mean all-reduce of the arena's live gradient range.  Use as `grad_hook` of RenderEngine.train_step."""

    def __init__(self, world_size: int, chunks: int = 1, group=None):
        self.world = world_size
        self.chunks = max(1, chunks)
        self.group = group

    def __call__(self, arena) -> None:
        if self.world <= 1:
            return
        lo, hi = arena.live_range
        flat = arena.grads[lo:hi]
        n = flat.numel()
        step = (n + self.chunks - 1) // self.chunks
        works = []
        for s in range(0, n, step):
            works.append(dist.all_reduce(flat[s : s + step], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()
        flat.mul_(1.0 / self.world)


def broadcast_params(arena, src: int = 0, group=None) -> None:
    """DDP's initial parameter broadcast from rank 0 (pipelines/base_pipeline.py:282)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(arena.params, src=src, group=group)


def rank_seed(base_seed: int, rank: int) -> int:
    """Every rank draws its own rays with seed + rank (scripts/train.py:97)."""
    return base_seed + rank
