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
