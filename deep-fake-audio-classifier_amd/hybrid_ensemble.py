"""Score-level ensembles on the HIP path -- counterparts of src/hybrid_ensemble.py (CNN2D sigmoid + CAE MSE, alpha
sweep), src/predict_hybrid.py (fixed-alpha prediction file) and src/ensemble.py (mean of sigmoids over checkpoints).

GPU side: every model scores the SAME resident stored-layout batch ([b,180,321], loaded once per rank): CNN2D and CNN1D
read it through the transposed view, the auto-encoder reads it raw with the FeatureNormalizer z-score fused into its
kernels and returns the per-sample MSE directly (no normalised copy, no reconstruction tensor).  Utterances are sharded
contiguously over ranks with no data-path collective; the N-float score vectors are gathered and fused on the host with
the reference's formulas (fusion.py).
"""
from __future__ import annotations

import argparse

import numpy as np
import pandas as pd
import torch

from . import distributed as dfa_dist
from . import fusion
from .dataloaders import FlatBatcher
from .dataset_cae import FeatureNormalizer
from .evaluation import calculate_eer
from .model import CNN2D
from .model_cae import ConvAutoencoder
from .model_cnn1d import CNN1D
from .predict import load_weights


@torch.no_grad()
def score_models(stored_features: torch.Tensor, cnn2d=None, cnn1d=None, cae=None, normalizer=None, batch_size=256,
                 device="cuda", rank=0, world=1):
    """One pass over this rank's shard; returns dict of numpy score vectors for the models that were given:
    'cnn2d' / 'cnn1d' sigmoid scores (src/hybrid_ensemble.py:31-43, src/ensemble.py:52-62), 'cae' per-sample MSE
    (src/hybrid_ensemble.py:46-61)."""
    for m in (cnn2d, cnn1d, cae):
        if m is not None:
            m.eval()
    mean = std = None
    if cae is not None and normalizer is not None:
        mean, std = normalizer.mean.to(device), normalizer.std.to(device)
    out = {k: [] for k, m in (("cnn2d", cnn2d), ("cnn1d", cnn1d), ("cae", cae)) if m is not None}
    for feats, _ in FlatBatcher(stored_features, None, batch_size, device=device, rank=rank, world=world):
        x = feats.transpose(1, 2)                                    # the strided [b,T,F] view, no copy
        if cnn2d is not None:
            out["cnn2d"].append(torch.sigmoid(cnn2d(x).squeeze(-1)))
        if cnn1d is not None:
            out["cnn1d"].append(torch.sigmoid(cnn1d(x).squeeze(-1)))
        if cae is not None:
            out["cae"].append(cae.score(x, mean, std))
    return {k: (torch.cat(v).double().cpu().numpy() if v else np.zeros(0)) for k, v in out.items()}


def hybrid_report(sup_scores, cae_scores, labels, alpha_steps=21):
    """EERs of the two score vectors alone and of their alpha mix (src/hybrid_ensemble.py:125-160)."""
    labels = list(labels)
    sup_eer, _ = calculate_eer(list(sup_scores), labels)
    cae_eer, _ = calculate_eer(list(cae_scores), labels)
    table, best_eer, best_alpha = fusion.alpha_sweep(sup_scores, cae_scores, labels, alpha_steps)
    return {"sup_eer": sup_eer, "cae_eer": cae_eer, "table": table, "best_eer": best_eer, "best_alpha": best_alpha}


def _stack(df):
    return torch.stack([f.float() for f in df["features"]]) if len(df) else torch.empty(0, 180, 321)


def _load(cls, path, device, **kw):
    return load_weights(cls(**kw).to(device), path, device)


def main(argv=None):
    p = argparse.ArgumentParser(description="Hybrid ensemble: supervised CNN2D + CAE anomaly scoring (MI355X HIP path).")
    p.add_argument("--sup-checkpoint", required=True)
    p.add_argument("--sup-arch", default="cnn2d", choices=["cnn2d"])
    p.add_argument("--cae-checkpoint", required=True)
    p.add_argument("--cae-normalizer", required=True)
    p.add_argument("--cnn1d-checkpoint", default=None, help="optional third member (mean-of-sigmoids with CNN2D)")
    p.add_argument("--dev-features", default="data/dev/features.pkl")
    p.add_argument("--dev-labels", default="data/dev/labels.pkl")
    p.add_argument("--batch-size", type=int, default=256)
    p.add_argument("--device", default="cuda")
    p.add_argument("--alpha-steps", type=int, default=21)
    p.add_argument("--alpha", type=float, default=None, help="write --out with this fixed alpha (predict_hybrid.py)")
    p.add_argument("--out", default=None)
    p.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    args = p.parse_args(argv)

    rank, world = dfa_dist.init()
    table = pd.merge(pd.read_pickle(args.dev_features), pd.read_pickle(args.dev_labels), on="uttid", how="inner") \
        .reset_index(drop=True)
    sup = _load(CNN2D, args.sup_checkpoint, args.device, in_features=180, dropout=0.2, precision=args.precision)
    cae = _load(ConvAutoencoder, args.cae_checkpoint, args.device, precision=args.precision)
    c1d = _load(CNN1D, args.cnn1d_checkpoint, args.device) if args.cnn1d_checkpoint else None
    norm = FeatureNormalizer.load(args.cae_normalizer)
    local = score_models(_stack(table), sup, c1d, cae, norm, args.batch_size, args.device, rank, world)
    scores = {k: dfa_dist.gather_scores(v) for k, v in local.items()}
    if rank != 0:
        return
    labels = table["label"].tolist()
    sup_scores = scores["cnn2d"] if c1d is None else fusion.ensemble_mean([scores["cnn2d"], scores["cnn1d"]])
    rep = hybrid_report(sup_scores, scores["cae"], labels, args.alpha_steps)
    print(f"Supervised-only  EER = {rep['sup_eer']:.6f}")
    print(f"CAE-only         EER = {rep['cae_eer']:.6f}")
    for a, e in rep["table"]:
        print(f"  {a:.2f}    {e:.6f}")
    print(f"Best hybrid EER:     {rep['best_eer']:.6f}  (alpha={rep['best_alpha']:.2f})")
    if args.out:
        alpha = rep["best_alpha"] if args.alpha is None else args.alpha
        pd.DataFrame({"uttid": table["uttid"].values,
                      "predictions": fusion.hybrid_scores(sup_scores, scores["cae"], alpha)}).to_pickle(args.out)


if __name__ == "__main__":
    main()
