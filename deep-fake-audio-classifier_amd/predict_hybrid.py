"""Fixed-alpha hybrid prediction file -- counterpart of src/predict_hybrid.py (same flags, same outputs).

    python -m dfa_amd.predict_hybrid --sup-checkpoint cnn2d_best.pt --cae-checkpoint cae_best.pt \\
           --cae-normalizer normalizer.pt --test-features features.pkl [--alpha 0.8] [--out prediction_hybrid.pkl]

Scores come from the HIP path: the CNN2D and the auto-encoder score the SAME resident stored-layout batch (the auto-encoder
with the FeatureNormalizer z-score and the per-sample MSE fused into its kernels; hybrid_ensemble.score_models); the
min-max normalisation and the alpha mix are the reference's formulas on the host (src/predict_hybrid.py:81-85,149-151).
The prediction file has the reference's schema (uttid, predictions float64), so scripts/evaluation.py reads it unchanged;
the optional comparison with an existing submission prints the reference's report (src/predict_hybrid.py:171-207)."""
from __future__ import annotations

import argparse
import pickle

import numpy as np
import pandas as pd
import torch

from . import distributed as dfa_dist
from . import fusion
from .dataset_cae import FeatureNormalizer
from .hybrid_ensemble import _load, _stack, score_models
from .model import CNN2D
from .model_cae import ConvAutoencoder


def print_distribution(name, scores):
    """src/predict_hybrid.py:88-95."""
    scores = np.asarray(scores)
    print(f"\n  {name}")
    print(f"    min={scores.min():.6f}  max={scores.max():.6f}")
    print(f"    mean={scores.mean():.6f}  median={np.median(scores):.6f}")
    print(f"    std={scores.std():.6f}")
    print(f"    est real (>0.5): {(scores > 0.5).sum()}  est fake (<=0.5): {(scores <= 0.5).sum()}")


def compare_with_submission(pred_df, path):
    """Per-sample difference and class agreement against an existing submission (a DataFrame pickle, or the course's
    submission dict whose 'predictions' entry is the DataFrame) -- src/predict_hybrid.py:171-207."""
    with open(path, "rb") as f:
        existing = pickle.load(f)
    old_df = existing["predictions"] if isinstance(existing, dict) and "predictions" in existing else existing
    print_distribution("Existing submission", old_df["predictions"].values)
    merged = pd.merge(pred_df, old_df, on="uttid", suffixes=("_new", "_old"))
    diff = merged["predictions_new"].values - merged["predictions_old"].values
    print("\n  Per-sample diff (new - old):")
    print(f"    mean={diff.mean():.6f}  std={diff.std():.6f}")
    print(f"    min={diff.min():.6f}  max={diff.max():.6f}")
    new_class = (merged["predictions_new"].values > 0.5).astype(int)
    old_class = (merged["predictions_old"].values > 0.5).astype(int)
    agree = int((new_class == old_class).sum())
    print(f"    class agreement: {agree}/{len(merged)} ({100 * agree / max(len(merged), 1):.1f}%)")
    bad = np.where(new_class != old_class)[0]
    for i in bad[:10 if len(bad) > 20 else 20]:
        row = merged.iloc[i]
        print(f"      {row['uttid']}: old={row['predictions_old']:.4f} new={row['predictions_new']:.4f}")
    return {"agree": agree, "n": len(merged), "mean_diff": float(diff.mean()) if len(diff) else 0.0}


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Hybrid (CNN2D + auto-encoder) predictions on a feature file (MI355X HIP path).")
    p.add_argument("--sup-checkpoint", required=True)
    p.add_argument("--cae-checkpoint", required=True)
    p.add_argument("--cae-normalizer", required=True)
    p.add_argument("--test-features", required=True, help="Path to final test features.pkl")
    p.add_argument("--existing-submission", default=None, help="Path to existing .pkl submission for comparison")
    p.add_argument("--alpha", type=float, default=0.80, help="Hybrid weight: alpha*supervised + (1-alpha)*cae")
    p.add_argument("--out", default="prediction_hybrid.pkl", help="Output prediction pkl")
    p.add_argument("--batch-size", type=int, default=32)
    p.add_argument("--device", default=None)
    p.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    device = args.device or "cuda"                              # src/predict_hybrid.py:17-24; this path has no CPU fallback
    rank, world = dfa_dist.init()
    features_df = pd.read_pickle(args.test_features)
    if "uttid" not in features_df.columns:
        raise ValueError("features.pkl must contain 'uttid'")
    if rank == 0:
        print(f"Test set: {len(features_df)} samples")
    sup = _load(CNN2D, args.sup_checkpoint, device, in_features=180, dropout=0.2, precision=args.precision)
    cae = _load(ConvAutoencoder, args.cae_checkpoint, device, precision=args.precision)
    norm = FeatureNormalizer.load(args.cae_normalizer)
    local = score_models(_stack(features_df), sup, None, cae, norm, args.batch_size, device, rank, world)
    dev = torch.device(device) if str(device).startswith("cuda") else None
    scores = {k: dfa_dist.gather_scores(v, device=dev) for k, v in local.items()}
    if rank != 0:
        return None
    hybrid = fusion.hybrid_scores(scores["cnn2d"], scores["cae"], args.alpha)       # src/predict_hybrid.py:149-151
    if len(hybrid) != len(features_df):
        raise ValueError(f"Prediction count mismatch: {len(hybrid)} vs {len(features_df)}")
    pred_df = pd.DataFrame({"uttid": features_df["uttid"].values, "predictions": hybrid.astype(np.float64)})
    pred_df.to_pickle(args.out)
    print(f"\nSaved hybrid predictions to {args.out}")
    print(f"\n{'=' * 60}\nDistribution Comparison")
    print_distribution("Supervised-only (sigmoid)", scores["cnn2d"])
    print_distribution("CAE-only (raw MSE, higher=real)", scores["cae"])
    print_distribution(f"Hybrid (alpha={args.alpha})", hybrid)
    if args.existing_submission:
        compare_with_submission(pred_df, args.existing_submission)
    print(f"\n{'=' * 60}")
    return pred_df


if __name__ == "__main__":
    main()
