"""Train-time feature augmentations on [B, T, F] batches -- counterparts of src/augmentation.py:5-186.

Semantics kept: one random draw per BATCH (not per sample) from Python's `random` for the shift / mask spans and from
the torch generator for the channel mask and the jitter noise, so seeded runs pick the same spans as the reference.
They are cheap, memory-bound device ops that run before the first HIP kernel reads the batch.
"""
from __future__ import annotations

import random

import torch


def time_shift(features: torch.Tensor, max_shift_ratio: float = 0.1) -> torch.Tensor:
    """Circular shift along T by a random amount in [-max_shift_ratio*T, +max_shift_ratio*T]."""
    if max_shift_ratio <= 0 or features.shape[1] <= 1:
        return features
    limit = int(features.shape[1] * max_shift_ratio)
    if limit < 1:
        return features
    shift = random.randint(-limit, limit)
    return features if shift == 0 else torch.roll(features, shifts=shift, dims=1)


def channel_drop(features: torch.Tensor, drop_prob: float = 0.1) -> torch.Tensor:
    """Zero whole feature dims with probability drop_prob (one [1,1,F] mask for the batch)."""
    if drop_prob <= 0:
        return features
    keep = (torch.rand((1, 1, features.shape[2]), device=features.device) >= drop_prob).to(features.dtype)
    return features * keep


def gaussian_jitter(features: torch.Tensor, std: float = 0.01) -> torch.Tensor:
    """Add N(0, std^2) noise."""
    if std <= 0:
        return features
    return features + torch.randn_like(features) * std


def compose(*fns):
    active = [fn for fn in fns if fn is not None]

    def _apply(x: torch.Tensor) -> torch.Tensor:
        for fn in active:
            x = fn(x)
        return x
    return _apply


def _span(n: int, lo_ratio: float, hi_ratio: float):
    length = max(1, min(int(n * random.uniform(lo_ratio, hi_ratio)), n - 1))
    start = random.randint(0, n - length)
    return start, length


def time_mask(features, max_mask_ratio=0.2, min_mask_ratio=0.05):
    """SpecAugment time masking: zero one contiguous span of frames (same span for the whole batch)."""
    start, length = _span(features.shape[1], min_mask_ratio, max_mask_ratio)
    out = features.clone()
    out[:, start:start + length, :] = 0
    return out


def feature_mask(features, max_mask_ratio=0.1, min_mask_ratio=0.02):
    """SpecAugment feature masking: zero one contiguous span of feature dims."""
    start, length = _span(features.shape[2], min_mask_ratio, max_mask_ratio)
    out = features.clone()
    out[:, :, start:start + length] = 0
    return out


def spec_augment(features, time_mask_ratio=0.2, feature_mask_ratio=0.1, apply_time_mask=True,
                 apply_feature_mask=False):
    if apply_time_mask:
        features = time_mask(features, max_mask_ratio=time_mask_ratio)
    if apply_feature_mask:
        features = feature_mask(features, max_mask_ratio=feature_mask_ratio)
    return features
