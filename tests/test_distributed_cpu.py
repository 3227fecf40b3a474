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


# ---- round 2: data-parallel robustness (unequal shard lengths hang the per-step all-reduce) --------------
def test_train_shard_indices_equal_steps_and_batches():
    from dfa_amd.dataloaders import train_shard_indices
    for n, world, bs in ((2050, 8, 256), (1025, 8, 32), (7, 2, 4), (11, 2, 4), (64, 4, 16), (3, 8, 2)):
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(n))
        shards = [train_shard_indices(perm, bs, r, world) for r in range(world)]
        steps = {len(s) // bs for s in shards}
        assert len(steps) == 1 and all(len(s) % bs == 0 for s in shards), (n, world, bs)   # same step count, full batches
        allidx = torch.cat(shards)
        assert set(allidx.tolist()) == set(range(n))                                       # every sample is seen
        assert len(allidx) - n < world * bs                                                # wrap-around pad < one global batch
        # global batch g = the ranks' g-th local batches = a contiguous run of the (padded) permutation
        g0 = torch.cat([s[:bs] for s in shards])
        assert torch.equal(g0, perm.repeat(-(-world * bs // n) + 1)[:world * bs])
    perm = torch.randperm(10)
    assert torch.equal(train_shard_indices(perm, 4, 0, 1), perm)                           # world 1: ragged tail kept


class _StubClassifier(torch.nn.Module):
    """CPU stand-in with the reference's call contract (x[B,T,F] -> [B,1]); the product model has no CPU path."""

    def __init__(self):
        super().__init__()
        self.bn = torch.nn.BatchNorm1d(3)
        self.lin = torch.nn.Linear(3, 1)

    def forward(self, x):
        return self.lin(self.bn(x.mean(dim=1)))


def _worker_dp(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import dfa_amd  # noqa: F401
    from dfa_amd import distributed as D
    from dfa_amd.dataloaders import FlatBatcher, train_shard_indices
    from dfa_amd.evaluation import evaluate_sharded
    D.init(backend="gloo")
    # N = 11 with world 2 x batch 4: ceil-sharding gave rank 0 two steps (6 samples) and rank 1 two steps (5) here, but e.g.
    # N = 9 gave 2 vs 1 -> mismatched all-reduce counts -> hang.  Equal-step shards: same count on every rank.
    for n in (9, 11):
        feats = torch.arange(n, dtype=torch.float32).view(n, 1, 1).expand(n, 3, 2).contiguous()
        labels = (torch.arange(n) % 2).float()
        perm = torch.randperm(n, generator=torch.Generator().manual_seed(3))
        idx = train_shard_indices(perm, 4, rank, world)
        steps = 0
        for fb, lb in FlatBatcher(feats[idx], labels[idx], 4, device="cpu"):
            assert fb.shape[0] == 4                              # equal local batch sizes: the 1/world scale is exact
            g = torch.full((10,), float(fb[:, 0, 0].sum()))
            D.allreduce_flat_(g)                                 # one collective per step on every rank
            steps += 1
        counts = [None] * world
        dist.all_gather_object(counts, steps)
        assert len(set(counts)) == 1 and steps == -(-n // 8), counts
    # BatchNorm running statistics: averaged before evaluation
    torch.manual_seed(0)
    model = _StubClassifier()
    with torch.no_grad():
        model.bn.running_mean.fill_(float(rank))
        model.bn.running_var.fill_(1.0 + 2.0 * rank)
    D.average_tensors_(D.bn_running_stats(model))
    assert torch.allclose(model.bn.running_mean, torch.full((3,), 0.5)) and torch.allclose(model.bn.running_var, torch.full((3,), 2.0))
    assert D.mean_scalar(float(rank + 1)) == 1.5 and D.mean_scalar(None) is None
    assert D.rank_seed(7, 0) == 7 and D.rank_seed(7, 1) != D.rank_seed(7, 2) != 7
    # sharded dev evaluation: every rank gets the metrics of the WHOLE set, identical to the unsharded evaluation
    n = 13
    g = torch.Generator().manual_seed(1)
    dev = torch.randn(n, 3, 5, generator=g)                      # stored layout [N, F, T]
    y = (torch.rand(n, generator=g) > 0.5).float()
    crit = torch.nn.BCEWithLogitsLoss()
    m_sh, s_sh, l_sh = evaluate_sharded(model, dev, y, criterion=crit, device="cpu", swap_tf=True, batch_size=4, rank=rank, world=world)
    model.eval()
    with torch.no_grad():
        want = model(dev.transpose(1, 2)).squeeze(-1).double().numpy()
    assert np.allclose(np.asarray(s_sh), want, atol=1e-6) and l_sh == y.tolist()
    both = [None] * world
    dist.all_gather_object(both, (m_sh["eer"], m_sh["avg_loss"], m_sh["threshold"]))
    assert both[0] == both[1]                                    # same decision inputs on every rank
    assert abs(m_sh["avg_loss"] - float(crit(torch.from_numpy(want).float(), y))) < 1e-6
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"dp_ok{rank}"), "w").write("ok")


@pytest.mark.timeout(300)
def test_world_size_2_data_parallel_robustness(tmp_path):
    world, port = 2, 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_dp, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"dp_ok{r}") for r in range(world))


def _worker_flat_input(rank, world, port, tmp):
    """Round 3: the data-parallel INPUT path.  A features.pkl is converted once, by rank 0 alone; every rank maps
    the flat file and fetches only the rows of its share of each global batch; the dev scores travel as a tensor all-gather."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import pandas as pd
    import dfa_amd  # noqa: F401
    from dfa_amd import dataset as ds_mod, distributed as D
    from dfa_amd.dataloaders import FlatBatcher, IndexedFlatBatcher, open_flat, train_shard_indices
    D.init(backend="gloo")
    n, F, T, bs = 37, 6, 5, 4
    fpath, lpath = os.path.join(tmp, "features.pkl"), os.path.join(tmp, "labels.pkl")
    if rank == 0:
        g = torch.Generator().manual_seed(0)
        feats = [torch.full((F, T), float(i)) + 0.001 * torch.randn(F, T, generator=g) for i in range(n)]
        pd.DataFrame({"uttid": [f"u{i:03d}" for i in range(n)], "features": feats}).to_pickle(fpath)
        pd.DataFrame({"uttid": [f"u{i:03d}" for i in range(n)], "label": [i % 2 for i in range(n)]}).to_pickle(lpath)
    dist.barrier()
    # nobody but rank 0 may un-pickle the features: make any other rank's pandas reader fail loudly
    real_read = pd.read_pickle
    if rank != 0:
        def deny(path, *a, **k):
            raise AssertionError(f"rank {rank} un-pickled {path}")
        pd.read_pickle = deny
    stacked_calls = []
    real_stacked = ds_mod.AudioDeepfakeDataset.stacked
    ds_mod.AudioDeepfakeDataset.stacked = lambda self, *a, **k: (stacked_calls.append(1), real_stacked(self, *a, **k))[1]
    feats, labels, uttids = open_flat(fpath, lpath, os.path.join(tmp, "cache"), rank, world, "train")
    pd.read_pickle = real_read
    assert not stacked_calls and tuple(feats.shape) == (n, F, T) and len(uttids) == n
    assert isinstance(feats.numpy().base, np.memmap) or not feats.numpy().flags.owndata      # a view of the mapped file, not a copy
    # a flat prefix is opened directly, by every rank, with no conversion and no pickle at all
    feats2, _, _ = open_flat(os.path.join(tmp, "cache", "train_flat"), None, os.path.join(tmp, "unused"), rank, world, "x")
    assert torch.equal(feats2, feats) and not os.path.exists(os.path.join(tmp, "unused"))
    # --- training epoch: same permutation on every rank, each rank fetches its rows only
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(5))
    idx = train_shard_indices(perm, bs, rank, world)
    batcher = IndexedFlatBatcher(feats, labels, idx, bs, device="cpu")
    seen, nb = [], 0
    for fb, lb in batcher:
        assert fb.shape[0] == bs and torch.equal(lb, labels[idx[nb * bs:(nb + 1) * bs]])
        seen.extend(int(round(float(v))) for v in fb[:, 0, 0])
        nb += 1
    assert seen == idx.tolist() and nb == len(batcher)
    row_bytes = F * T * 4
    total_rows = -(-n // (world * bs)) * world * bs                     # wrap-around padded to whole global batches
    assert batcher.rows_fetched == total_rows // world and batcher.bytes_fetched == batcher.rows_fetched * row_bytes
    # per-rank bytes pulled from the source ~ 1 / world of an epoch (never the whole set)
    assert batcher.bytes_fetched <= (n * row_bytes) / world + world * bs * row_bytes
    counts = torch.zeros(n)
    counts.index_add_(0, idx, torch.ones(idx.numel()))
    dist.all_reduce(counts)
    assert int(counts.min()) >= 1 and int(counts.sum()) == total_rows     # together the ranks cover every utterance
    # --- dev evaluation: contiguous shards + tensor all-gather of uneven score vectors (no pickled objects)
    real_ago = dist.all_gather_object
    dist.all_gather_object = lambda *a, **k: (_ for _ in ()).throw(AssertionError("all_gather_object on the per-epoch path"))
    lo, hi = D.shard_range(n, rank, world)
    local = [float(fb[i, 0, 0]) for fb, _ in FlatBatcher(feats, None, bs, device="cpu", rank=rank, world=world) for i in range(fb.shape[0])]
    assert len(local) == hi - lo
    allscores = D.gather_scores(local)
    dist.all_gather_object = real_ago
    assert allscores.shape == (n,) and np.allclose(allscores, feats[:, 0, 0].double().numpy())
    assert np.array_equal(D.gather_scores([]) if False else allscores, allscores)
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"flat_ok{rank}"), "w").write("ok")


def test_world_size_2_flat_input_path(tmp_path):
    world, port = 2, 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_flat_input, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"flat_ok{r}") for r in range(world))


def _worker_cae_flat_input(rank, world, port, tmp):
    """Round 3: the auto-encoder's data-parallel input path (train_cae.py under torchrun; src/train_cae.py:108-160 builds the
    normaliser and the bonafide datasets on one GPU).  Rank 0 alone converts the pickles; the normaliser is fitted from each
    rank's share of the bonafide rows + one all-reduce and equals FeatureNormalizer.fit on the whole bonafide set; a rank's
    batches are its rows of the flat source, z-scored as the reference's dataset does."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import pandas as pd
    import dfa_amd  # noqa: F401
    from dfa_amd import distributed as D
    from dfa_amd.dataloaders import IndexedFlatBatcher, open_flat, train_shard_indices
    from dfa_amd.dataset_cae import FeatureNormalizer, fit_normalizer_sharded
    from dfa_amd.train_cae import _NormalizedBatches
    D.init(backend="gloo")
    n, F, T, bs = 41, 8, 7, 3
    fpath, lpath = os.path.join(tmp, "features.pkl"), os.path.join(tmp, "labels.pkl")
    g = torch.Generator().manual_seed(3)
    feats_list = [torch.randn(F, T, generator=g) * (1.0 + 0.1 * i) + 0.3 * i for i in range(n)]      # same on every rank (seeded)
    labels_list = [1 if (i % 3) else 0 for i in range(n)]
    if rank == 0:
        pd.DataFrame({"uttid": [f"u{i:03d}" for i in range(n)], "features": feats_list}).to_pickle(fpath)
        pd.DataFrame({"uttid": [f"u{i:03d}" for i in range(n)], "label": labels_list}).to_pickle(lpath)
    dist.barrier()
    real_read = pd.read_pickle
    if rank != 0:
        def deny(path, *a, **k):
            raise AssertionError(f"rank {rank} un-pickled {path}")
        pd.read_pickle = deny
    feats, labels, _ = open_flat(fpath, lpath, os.path.join(tmp, "cache"), rank, world, "cae_train")
    pd.read_pickle = real_read
    rows = (labels == 1).nonzero().reshape(-1)
    assert rows.tolist() == [i for i in range(n) if labels_list[i] == 1]
    norm = fit_normalizer_sharded(feats, rows, rank, world)
    want = FeatureNormalizer().fit([feats_list[i].transpose(0, 1) for i in rows.tolist()])           # the reference's fit, whole set
    assert torch.allclose(norm.mean, want.mean, rtol=1e-5, atol=1e-6) and torch.allclose(norm.std, want.std, rtol=1e-5, atol=1e-6)
    assert norm.rows_fetched <= -(-rows.numel() // world)                                            # its share, not the set
    both = [torch.zeros(2, F) for _ in range(world)]
    dist.all_gather(both, torch.stack([norm.mean, norm.std]))
    assert all(torch.equal(b, both[0]) for b in both)                                                # identical on every rank
    perm = torch.randperm(rows.numel(), generator=torch.Generator().manual_seed(9))
    idx = rows[train_shard_indices(perm, bs, rank, world)]
    batcher = IndexedFlatBatcher(feats, None, idx, bs, device="cpu")
    k = 0
    for x in _NormalizedBatches(batcher, norm.mean, norm.std):
        assert tuple(x.shape) == (bs, T, F)
        for j in range(bs):
            ref = (feats_list[int(idx[k])].transpose(0, 1) - norm.mean) / norm.std
            assert torch.allclose(x[j], ref, rtol=1e-6, atol=1e-6)
            k += 1
    assert k == idx.numel() and batcher.rows_fetched == idx.numel()
    open(os.path.join(tmp, f"cae_flat_ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def test_world_size_2_cae_flat_input_path(tmp_path):
    world, port = 2, 33500 + (os.getpid() % 2000)
    mp.spawn(_worker_cae_flat_input, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"cae_flat_ok{r}") for r in range(world))
