"""Anomaly scoring with the auto-encoder -- counterpart of src/evaluation_cae.py:30-87.

evaluate_cae(model, dataloader, device) keeps the reference contract (loader yields normalised (features, label)
batches) and returns (metrics, mse_scores, labels) with both score polarities tried, exactly as the reference does.
evaluate_cae_raw(...) is the GPU-sized variant: raw stored-layout features + normaliser statistics, z-score and
per-sample MSE fused into the kernels.
"""
from __future__ import annotations

import numpy as np
import torch

from .evaluation import calculate_eer


def _metrics(all_mse: np.ndarray, labels: np.ndarray) -> dict:
    eer_neg, thr_neg = calculate_eer((-all_mse).tolist(), labels.tolist())    # fakes have MORE error
    eer_pos, thr_pos = calculate_eer(all_mse.tolist(), labels.tolist())       # fakes have LESS error
    if eer_neg <= eer_pos:
        eer, thr, conv = eer_neg, -thr_neg, "standard (-MSE: fakes have higher error)"
    else:
        eer, thr, conv = eer_pos, thr_pos, "inverted (+MSE: fakes have lower error)"
    return {"avg_mse": float(np.mean(all_mse)), "avg_mse_bonafide": float(np.mean(all_mse[labels == 1])),
            "avg_mse_spoof": float(np.mean(all_mse[labels == 0])), "eer": eer, "eer_neg": eer_neg, "eer_pos": eer_pos,
            "threshold_mse": thr, "convention": conv}


@torch.no_grad()
def evaluate_cae(model, dataloader, device="cuda"):
    model.eval()
    chunks, labels = [], []
    for features, batch_labels in dataloader:
        chunks.append(model.score(features.to(device, non_blocking=True)))
        labels.extend(batch_labels.tolist())
    all_mse = torch.cat(chunks).cpu().numpy().astype(np.float64) if chunks else np.zeros(0)
    labels = np.array(labels)
    return _metrics(all_mse, labels), all_mse, labels


@torch.no_grad()
def evaluate_cae_raw(model, stored_features, labels, normalizer, batch_size=256, device="cuda", rank=0, world=1):
    """stored_features [N,180,321] (CPU or GPU), labels [N]; the z-score is fused into the kernels."""
    from .dataloaders import FlatBatcher
    model.eval()
    mean, std = normalizer.mean.to(device), normalizer.std.to(device)
    chunks = []
    for feats, _ in FlatBatcher(stored_features, None, batch_size, device=device, rank=rank, world=world):
        chunks.append(model.score(feats.transpose(1, 2), mean, std))
    all_mse = torch.cat(chunks).cpu().numpy().astype(np.float64)
    if world > 1:
        return all_mse
    labels = np.asarray(labels)
    return _metrics(all_mse, labels), all_mse, labels
