"""The synthetic EER-parity set of SURVEY.md section 8(d): N utterances in the reference's features.pkl / labels.pkl
schema (DataFrame cols `uttid` str + `features` torch.FloatTensor [180, 321]; `uttid` + `label` int 0/1 --
README.md:45-103, src/dataset.py:24-30), 45 % label 1, class 1 = class 0 + a fixed low-rank pattern with a per-utterance
strength, scaled so that the golden CNN2D weights (tests/golden/cnn2d_eval.npz) score it with an EER strictly inside
(0, 5 %): the metric is then sensitive to single rank swaps near the threshold.

The pattern is the rank-4 truncation of d(logit)/d(x) of the REFERENCE model averaged over 16 random utterances
(computed by tests/golden/make_golden_r2.py with the reference's own autograd, committed in tests/golden/eer2000.npz
next to the reference's logits, sigmoid scores and EERs on the set).  The GPU parity test regenerates the same features
from the same seeds (torch CPU generator: bit-reproducible for a fixed torch build) and checks a checksum first."""
import numpy as np
import torch

N_DEFAULT = 2000
SEED = 20260
PATTERN_SCALE = 0.2           # calibrated with the golden weights: EER of a few per cent (make_golden_r2.py prints it)
STRENGTH_JITTER = 0.45
F, T = 180, 321


def make_eer_set(pattern, n=N_DEFAULT, seed=SEED, scale=PATTERN_SCALE, jitter=STRENGTH_JITTER):
    """pattern: [F, T] float32 -> (features [n, F, T] float32 in the stored layout, labels [n] int64 with ~45 % ones,
    uttids list[str])."""
    pattern = torch.as_tensor(pattern, dtype=torch.float32)
    g = torch.Generator().manual_seed(seed)
    labels = (torch.rand(n, generator=g) < 0.45).long()
    feats = torch.randn(n, F, T, generator=g) * 3.2 - 0.07
    strength = scale * (1.0 + jitter * torch.randn(n, generator=g))      # the two score distributions overlap: EER > 0
    feats += (labels.float() * strength)[:, None, None] * pattern[None]
    feats.clamp_(-61.0, 87.0)
    uttids = [f"dev_{i:06d}" for i in range(n)]
    return feats, labels, uttids


def checksum(feats):
    """order-sensitive float64 checksum of the feature tensor (detects RNG drift between torch builds)."""
    v = feats.double().reshape(-1)
    w = (torch.arange(1, v.numel() + 1, dtype=torch.float64) % 9973) + 1.0
    return float((v * w).sum())


def to_frames(feats, labels, uttids):
    import pandas as pd
    # .clone(): a VIEW pickles its whole backing storage (2000 x 462 MB); every row owns its [180, 321] tensor
    fdf = pd.DataFrame({"uttid": uttids, "features": [feats[i].clone() for i in range(feats.shape[0])]})
    ldf = pd.DataFrame({"uttid": uttids, "label": labels.numpy().astype(np.int64)})
    return fdf, ldf
