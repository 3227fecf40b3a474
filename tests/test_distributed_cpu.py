"""World-size-2 tests of the data-parallel plumbing on CPU (gloo): utterance sharding + score gathering, and the
single flat-gradient all-reduce + broadcast used by the training step."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import dfa_amd  # noqa: F401
    from dfa_amd import distributed as D
    from dfa_amd.dataloaders import FlatBatcher
    r, w = D.init(backend="gloo")
    assert (r, w) == (rank, world)
    # --- inference: contiguous utterance shards, no collective on the data path, ordered gather of the scores
    n = 11
    feats = torch.arange(n, dtype=torch.float32).view(n, 1, 1).expand(n, 3, 2).contiguous()
    lo, hi = D.shard_range(n, rank, world)
    local = []
    for fb, _ in FlatBatcher(feats, None, 4, device="cpu", rank=rank, world=world):
        local.extend(fb[:, 0, 0].tolist())                      # "score" = utterance id
    assert local == list(range(lo, hi))
    allscores = D.gather_scores(local)
    assert allscores.tolist() == list(range(n))
    # --- training: one flat gradient buffer, SUM all-reduce, 1/world folded into the optimiser step
    g = torch.full((1000,), float(rank + 1))
    D.allreduce_flat_(g)
    assert torch.all(g == sum(range(1, world + 1)))
    mean_grad = g / world
    assert torch.allclose(mean_grad, torch.full((1000,), (world + 1) / 2))
    p = torch.full((7,), float(rank))
    D.broadcast_parameters_(p, src=0)
    assert torch.all(p == 0)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_world_size_2_gloo(tmp_path):
    world, port = 2, 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_shard_range_covers_everything():
    from dfa_amd.distributed import shard_range
    for n in (0, 1, 7, 8, 2000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            covered = [i for lo, hi in spans for i in range(lo, hi)]
            assert covered == list(range(n))


def test_augmentations_semantics():
    import random
    from dfa_amd import augmentation as A
    x = torch.arange(2 * 20 * 6, dtype=torch.float32).view(2, 20, 6) + 1
    random.seed(0)
    y = A.time_mask(x, max_mask_ratio=0.3, min_mask_ratio=0.1)
    rows = (y == 0).all(dim=2).all(dim=0)
    assert 1 <= int(rows.sum()) <= 6 and y.shape == x.shape                      # one contiguous span, whole batch
    idx = rows.nonzero().flatten()
    assert int(idx[-1] - idx[0]) + 1 == int(rows.sum())
    y = A.feature_mask(x, max_mask_ratio=0.5, min_mask_ratio=0.2)
    cols = (y == 0).all(dim=1).all(dim=0)
    assert 1 <= int(cols.sum()) <= 3
    random.seed(1)
    y = A.time_shift(x, max_shift_ratio=0.2)
    assert torch.equal(torch.sort(y[0, :, 0]).values, x[0, :, 0])                # a roll: same frames, rotated
    torch.manual_seed(0)
    y = A.channel_drop(x, drop_prob=0.5)
    kept = (y != 0).any(dim=1).any(dim=0)
    assert torch.equal(y[:, :, kept], x[:, :, kept]) and 0 < int(kept.sum()) < 6
    assert A.gaussian_jitter(x, std=0.0) is x and A.time_shift(x, 0.0) is x
    f = A.compose(None, lambda t: t + 1, None, lambda t: t * 2)
    assert torch.equal(f(x), (x + 1) * 2)
    with pytest.raises(ValueError):
        from dfa_amd.train import make_criterion
        make_criterion(0.5)
