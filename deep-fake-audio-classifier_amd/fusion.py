"""Score-level fusion on the host -- counterparts of src/hybrid_ensemble.py:64-69,139-151,
src/predict_hybrid.py:81-85,149-151 and src/ensemble.py:121.  Inputs are per-utterance score vectors produced on
the GPU (CNN2D sigmoid, CNN1D sigmoid, CAE per-sample MSE); everything here is O(N) numpy, as in the reference."""
from __future__ import annotations

import numpy as np

from .evaluation import calculate_eer


def normalise_scores(scores: np.ndarray) -> np.ndarray:
    """Min-max to [0, 1]; a (near-)constant vector maps to zeros (range < 1e-12)."""
    scores = np.asarray(scores)
    lo, hi = scores.min(), scores.max()
    if hi - lo < 1e-12:
        return np.zeros_like(scores)
    return (scores - lo) / (hi - lo)


def hybrid_scores(sup_scores, cae_scores, alpha: float) -> np.ndarray:
    """alpha * norm(supervised) + (1 - alpha) * norm(CAE MSE)   (alpha = 1 -> supervised only)."""
    return alpha * normalise_scores(np.asarray(sup_scores)) + (1 - alpha) * normalise_scores(np.asarray(cae_scores))


def alpha_sweep(sup_scores, cae_scores, labels, alpha_steps: int = 21):
    """EER for alpha in linspace(0, 1, alpha_steps); first strict minimum wins.  -> (table, best_eer, best_alpha)."""
    sup_n, cae_n = normalise_scores(np.asarray(sup_scores)), normalise_scores(np.asarray(cae_scores))
    labels = list(labels)
    table, best_eer, best_alpha = [], 1.0, 0.0
    for alpha in np.linspace(0.0, 1.0, alpha_steps):
        eer, _ = calculate_eer((alpha * sup_n + (1 - alpha) * cae_n).tolist(), labels)
        table.append((float(alpha), eer))
        if eer < best_eer:
            best_eer, best_alpha = eer, float(alpha)
    return table, best_eer, best_alpha


def ensemble_mean(score_vectors) -> np.ndarray:
    """Simple mean of per-model sigmoid scores (src/ensemble.py:121)."""
    return np.mean([np.asarray(s) for s in score_vectors], axis=0)


def gather_sharded(local_scores, world: int = 1):
    """Concatenate per-rank score vectors in rank order (contiguous utterance shards).  With world == 1 it is the
    identity; with world > 1 it uses torch.distributed.all_gather_object (scores are tiny: N floats)."""
    if world == 1:
        return np.asarray(local_scores)
    import torch.distributed as dist
    parts = [None] * world
    dist.all_gather_object(parts, np.asarray(local_scores))
    return np.concatenate(parts)
