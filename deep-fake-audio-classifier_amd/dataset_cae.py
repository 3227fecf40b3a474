"""Datasets and normaliser for the auto-encoder -- counterpart of src/dataset_cae.py:18-141.

FeatureNormalizer: per-feature-dim z-score (mean / UNBIASED std over all frames of the bonafide training set, std
clamped to >= 1e-8).  BonafideDataset / FullLabeledDataset keep the reference behaviour (items are (T, F) tensors,
normalised inside __getitem__).  On the GPU path the normalisation can instead be fused into the CAE kernels'
loads: pass `normalizer.mean/std` to `ConvAutoencoder.score(raw_view, mean, std)` and skip the normalised copy.
"""
from __future__ import annotations

import pandas as pd
import torch
from torch.utils.data import Dataset


class FeatureNormalizer:
    def __init__(self):
        self.mean: torch.Tensor | None = None   # (F,)
        self.std: torch.Tensor | None = None    # (F,)

    def fit(self, features_list):
        """mean/std over the concatenated frames of a list of (T, F) tensors."""
        frames = torch.cat(list(features_list), dim=0)
        self.mean = frames.mean(dim=0)
        self.std = frames.std(dim=0).clamp(min=1e-8)
        return self

    def transform(self, x: torch.Tensor) -> torch.Tensor:
        if self.mean is None:
            raise RuntimeError("Call .fit() first")
        return (x - self.mean.to(x.device)) / self.std.to(x.device)

    def save(self, path: str) -> None:
        torch.save({"mean": self.mean, "std": self.std}, path)

    @classmethod
    def load(cls, path: str) -> "FeatureNormalizer":
        blob = torch.load(path, map_location="cpu")
        obj = cls()
        obj.mean, obj.std = blob["mean"], blob["std"]
        return obj


def fit_normalizer_sharded(features: torch.Tensor, rows: torch.Tensor, rank: int = 0, world: int = 1, chunk: int = 64) -> FeatureNormalizer:
    """FeatureNormalizer.fit (src/dataset_cae.py:24-29: mean / unbiased std over all frames of the bonafide utterances, per
    feature dim) for the data-parallel driver: `features` is the flat [N, F, T] source (memory-mapped), `rows` the bonafide row
    indices.  Every rank sums ITS contiguous share of the rows in float64 (rows fetched: 1/world of them), one all-reduce of
    3 x F doubles makes the statistics identical everywhere -- nobody reads the whole set, nobody un-pickles."""
    import torch.distributed as dist
    rows = rows.to(torch.int64).reshape(-1)
    n = rows.numel()
    per = -(-n // world) if world > 0 else n
    mine = rows[rank * per: min(n, (rank + 1) * per)]
    F = features.shape[1]
    acc = torch.zeros(3, F, dtype=torch.float64)            # sum, sum of squares, frame count
    for lo in range(0, mine.numel(), chunk):
        blk = features[mine[lo:lo + chunk]].double()         # [b, F, T]
        acc[0] += blk.sum(dim=(0, 2))
        acc[1] += (blk * blk).sum(dim=(0, 2))
        acc[2] += blk.shape[0] * blk.shape[2]
    if world > 1 and dist.is_available() and dist.is_initialized():
        t = acc.cuda() if dist.get_backend() == "nccl" else acc
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        acc = t.cpu()
    cnt = acc[2]
    mean = acc[0] / cnt
    var = (acc[1] - cnt * mean * mean) / (cnt - 1).clamp(min=1)
    obj = FeatureNormalizer()
    obj.mean = mean.float()
    obj.std = var.clamp(min=0).sqrt().float().clamp(min=1e-8)
    obj.rows_fetched = int(mine.numel())
    return obj


def _merged(features_path, labels_path):
    return pd.merge(pd.read_pickle(features_path), pd.read_pickle(labels_path), on="uttid", how="inner")


class _CaeBase(Dataset):
    def __init__(self, normalizer, swap_tf):
        self.normalizer, self.swap_tf = normalizer, swap_tf

    def _prep(self, feat):
        feat = feat.float()
        if self.swap_tf:
            feat = feat.transpose(0, 1)           # stored (180, 321) -> (321, 180)
        if self.normalizer is not None:
            feat = self.normalizer.transform(feat)
        return feat


class BonafideDataset(_CaeBase):
    """label == 1 samples only; items are single tensors (reconstruction needs no label)."""

    def __init__(self, features_path, labels_path, normalizer=None, swap_tf=True):
        super().__init__(normalizer, swap_tf)
        table = _merged(features_path, labels_path)
        self.features = table[table["label"] == 1].reset_index(drop=True)["features"].tolist()

    def __len__(self):
        return len(self.features)

    def __getitem__(self, idx):
        return self._prep(self.features[idx])


class FullLabeledDataset(_CaeBase):
    """Both classes; items are (features, label float32)."""

    def __init__(self, features_path, labels_path, normalizer=None, swap_tf=True):
        super().__init__(normalizer, swap_tf)
        table = _merged(features_path, labels_path).reset_index(drop=True)
        self.features = table["features"].tolist()
        self.labels = table["label"].tolist()

    def __len__(self):
        return len(self.features)

    def __getitem__(self, idx):
        return self._prep(self.features[idx]), torch.tensor(self.labels[idx], dtype=torch.float32)

    def stacked_raw(self):
        """([N,180,321] float32 stored-layout features, [N] labels) for the fused-normalisation GPU path."""
        return torch.stack([f.float() for f in self.features]), torch.tensor(self.labels, dtype=torch.float32)


def build_normalizer(features_path, labels_path, swap_tf=True) -> FeatureNormalizer:
    table = _merged(features_path, labels_path)
    feats = [(f.float().transpose(0, 1) if swap_tf else f.float()) for f in table[table["label"] == 1]["features"]]
    return FeatureNormalizer().fit(feats)
