"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL/xGMI (backend "nccl" on ROCm; "gloo"
on CPU for tests).  The hot path needs exactly two patterns:

  * inference shards by utterance with NO data-path collective: rank r scores the contiguous range
    [r*ceil(N/world), ...) and the host concatenates the per-rank score vectors in rank order;
  * training exchanges ONE flat fp32 gradient buffer per step (464,644 bytes for CNN2D): a sum all-reduce followed
    by the 1/world scale folded into the fused AdamW kernel.  At this size the ring is latency-bound, not per-link
    bandwidth-bound, so there is no bucketing: one call, one buffer.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def init(backend: str | None = None, device: torch.device | None = None):
    """Initialise the default process group from torchrun's environment (RANK/WORLD_SIZE/MASTER_*).
    Returns (rank, world).  A single process (no WORLD_SIZE) needs no group: (0, 1)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kwargs = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend=backend, **kwargs)
    return dist.get_rank(), dist.get_world_size()


def shard_range(n: int, rank: int, world: int):
    """Contiguous utterance range of `rank`: [lo, hi) with ceil(n/world) utterances per rank (last ranks may be short)."""
    per = -(-n // world)
    return min(rank * per, n), min((rank + 1) * per, n)


def allreduce_flat_(flat: torch.Tensor, group=None) -> torch.Tensor:
    """In-place SUM all-reduce of one flat buffer (the whole model's gradients).  No-op without a process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def gather_scores(local_scores, group=None) -> np.ndarray:
    """All ranks get the concatenation (rank order) of every rank's score vector; shards may have different lengths."""
    import torch.distributed as dist
    local = np.asarray(local_scores, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    parts = [None] * dist.get_world_size(group)
    dist.all_gather_object(parts, local, group=group)
    return np.concatenate(parts)


def broadcast_parameters_(flat_params: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """Make every rank start from rank `src`'s weights (one broadcast of the flat parameter buffer)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
    return flat_params
