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


def init(backend: str | None = None, device: torch.device | None = None, force_group: bool = False):
    """Initialise the default process group from torchrun's environment (RANK/WORLD_SIZE/MASTER_*).
    Returns (rank, world).  A single process (no WORLD_SIZE) needs no group: (0, 1), unless force_group asks for a
    one-rank group (used to exercise the RCCL code path on a one-GPU box).  Backend: the argument, else the environment
    variable DFA_DIST_BACKEND, else "nccl" (= RCCL on ROCm) when a GPU is present, else "gloo"."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not force_group:
        return 0, 1
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = backend or os.environ.get("DFA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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


def gather_scores(local_scores, group=None, device=None) -> np.ndarray:
    """All ranks get the concatenation (rank order) of every rank's score vector.  Shards may have different lengths: one tiny
    all-gather of the lengths, then ONE tensor all-gather of the scores padded to the longest shard (float64; over RCCL the
    tensors live on `device`) -- no pickling of python objects on the per-epoch path."""
    import torch.distributed as dist
    local = np.asarray(local_scores, dtype=np.float64).reshape(-1)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    dev = device if (dist.get_backend(group) == "nccl" and device is not None) else torch.device("cpu")
    if dist.get_backend(group) == "nccl" and device is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    lens = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([local.size], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(lens, mine, group=group)
    lens = lens.cpu().tolist()
    per = max(max(lens), 1)
    buf = torch.zeros(per, dtype=torch.float64, device=dev)
    buf[:local.size] = torch.from_numpy(local).to(dev)
    out = torch.empty(world * per, dtype=torch.float64, device=dev)
    dist.all_gather_into_tensor(out, buf, group=group)
    out = out.cpu().numpy().reshape(world, per)
    return np.concatenate([out[r, :lens[r]] for r in range(world)])


def broadcast_parameters_(flat_params: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """Make every rank start from rank `src`'s weights (one broadcast of the flat parameter buffer)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
    return flat_params


def average_tensors_(tensors, group=None):
    """Replace every tensor by its mean over ranks with ONE all-reduce of a flat copy (BatchNorm running statistics
    before a dev evaluation / checkpoint: they are updated from rank-local batches and drift apart otherwise)."""
    import torch.distributed as dist
    tensors = list(tensors)
    if not tensors or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return tensors
    flat = torch.cat([t.detach().reshape(-1).float() for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    with torch.no_grad():
        for t in tensors:
            k = t.numel()
            t.copy_(flat[off:off + k].view_as(t))
            off += k
    return tensors


def bn_running_stats(model):
    """The running_mean / running_var buffers of a dfa_amd model, in module order."""
    return [b for name, b in model.named_buffers() if name.endswith("running_mean") or name.endswith("running_var")]


def rank_seed(seed: int, rank: int) -> int:
    """A per-rank 64-bit Philox key derived from the run seed (dropout masks and jitter noise must differ between the
    ranks of a data-parallel step; the mask SPANS stay a per-global-batch draw like the reference's per-batch draw)."""
    return (int(seed) + 0x9E3779B97F4A7C15 * (int(rank) + 1)) & 0xFFFFFFFFFFFFFFFF if rank else int(seed) & 0xFFFFFFFFFFFFFFFF


def mean_scalar(value, device=None, group=None):
    """Mean over ranks of a python float (None stays None): the epoch's training loss, so that every rank holds the same
    number when it feeds the best-checkpoint tie-break."""
    import torch.distributed as dist
    if value is None or not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return value
    use_cuda = dist.get_backend(group) == "nccl"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if use_cuda else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()) / dist.get_world_size(group)


def cpu_budget() -> int:
    """CPUs this process may actually use: the affinity mask and the cgroup CPU quota, not the machine's count.  A GPU box reports
    128 logical CPUs to a job that owns 16 of them; torch sizes its intra-op pool from the former, and every host-side tensor op
    (copies, index kernels, reductions) then runs 8x oversubscribed -- measured 112 ms instead of 3 for a 59 MB batch gather
    (tools/gpu_loader_probe.py)."""
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as fh:
                parts = fh.read().split()
            if path.endswith("cpu.max"):
                if parts and parts[0] != "max":
                    n = min(n, max(1, int(parts[0]) // int(parts[1])))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                    period = int(fh.read().split()[0])
                if quota > 0:
                    n = min(n, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def limit_cpu_threads() -> int:
    """Cap torch's intra-op thread pool at cpu_budget() (never raises it); the command-line drivers call this first."""
    n = cpu_budget()
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()
