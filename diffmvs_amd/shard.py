"""Multi-GPU plumbing for the inference path: reference views (or whole scenes) shard across
one process per GPU with no data-path collective (the reference itself only has
single-GPU inference, test.py:101-111).  torch.distributed is used for the rendezvous, the
barrier around the timed region and the max-over-ranks reduction of the elapsed time only
(backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests)."""
from __future__ import annotations

import os
from typing import List, Sequence

import torch


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_items(n_items: int, rank: int, world: int) -> List[int]:
    """Static round-robin partition of work items (reference views / scenes): item i -> rank i % world.
    Every item is owned by exactly one rank; ranks differ by at most one item."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_items, world))


def shard_scenes(scenes: Sequence[str], rank: int, world: int) -> List[str]:
    """Scene-granular sharding (keeps a scene's depth maps on one worker for the later fusion step)."""
    return [scenes[i] for i in shard_items(len(scenes), rank, world)]


def init_distributed(backend: str, device=None):
    import torch.distributed as td
    if td.is_initialized():
        return td
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    td.init_process_group(backend, **kw)
    return td


def barrier_and_max(elapsed_s: float, device="cpu") -> float:
    """Whole-job time of a sharded run = slowest rank (the driver's SCALE contract)."""
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return elapsed_s
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def total_items(n_local: int, device="cpu") -> int:
    import torch.distributed as td
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return n_local
    t = torch.tensor([n_local], dtype=torch.int64, device=device)
    td.all_reduce(t, op=td.ReduceOp.SUM)
    return int(t.item())
