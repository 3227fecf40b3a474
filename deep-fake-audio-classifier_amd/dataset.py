"""Dataset over the reference's pickle schema -- counterpart of src/dataset.py:6-56.

features.pkl: pandas DataFrame with columns `uttid` (str) and `features` (torch.FloatTensor [180, 321], i.e.
[feature_dim, seq_len]); labels.pkl: `uttid`, `label` (0 = deepfake, 1 = real) (README.md:41-103).
Items are (features float32 [180,321], label float32 scalar), exactly as the reference returns them.

Beyond the reference: the merged table is also available as ONE contiguous [N,180,321] tensor (`.stacked()`), which is
what a 10^5 utt/s consumer needs -- per-item `.iloc` + default collate tops out around 10^3 utt/s.
"""
from __future__ import annotations

import pandas as pd
import torch
from torch.utils.data import Dataset


class AudioDeepfakeDataset(Dataset):
    def __init__(self, features_path, labels_path=None):
        feats = pd.read_pickle(features_path)
        if "uttid" not in feats.columns:
            raise ValueError("features.pkl must contain 'uttid'")
        self.features = feats
        if labels_path is None:
            self.labels = None
            table = feats
        else:
            self.labels = pd.read_pickle(labels_path)
            table = pd.merge(feats, self.labels, on="uttid", how="inner")   # src/dataset.py:28
        self.data = table.reset_index(drop=True)
        self._stack = None

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        row = self.data.iloc[idx]
        feat = row["features"].float()
        if self.labels is None:
            return feat
        return feat, torch.tensor(row["label"], dtype=torch.float32)

    # ---- bulk access ---------------------------------------------------------------------------------------
    def uttids(self):
        return self.data["uttid"].values

    def stacked(self, pin: bool = False):
        """(features [N,180,321] float32 contiguous, labels [N] float32 or None); built once and cached."""
        if self._stack is None:
            feats = torch.stack([f.float() for f in self.data["features"]])
            labels = None if self.labels is None else torch.tensor(self.data["label"].values, dtype=torch.float32)
            if pin and torch.cuda.is_available():
                feats = feats.pin_memory()
            self._stack = (feats, labels)
        return self._stack
