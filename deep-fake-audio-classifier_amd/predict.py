"""Checkpoint -> prediction.pkl -- counterpart of src/predict.py (same CLI flags, same output schema: DataFrame
{uttid, predictions} with Python-float scores, sigmoid applied by default), with the forward on the MI355X HIP path.

    python -m dfa_amd.predict --features features.pkl --checkpoint cnn2d_best.pt --model cnn2d --out prediction.pkl
"""
from __future__ import annotations

import argparse

import pandas as pd
import torch

from .model import CNN2D


def build_model(name: str, in_features: int = 180, dropout: float = 0.3, precision: str = "fp32"):
    if name == "cnn2d":
        return CNN2D(in_features=in_features, dropout=dropout, precision=precision)
    if name == "cnn1d":
        from .model_cnn1d import CNN1D
        return CNN1D(in_features=in_features, dropout=dropout)
    raise ValueError(f"unknown model {name!r} (choices: cnn2d, cnn1d)")


def load_weights(model, checkpoint_path: str, device="cuda"):
    """Accept the reference's checkpoint dict {"model_state": ...} or a bare state_dict (src/predict.py:78-85)."""
    try:
        ckpt = torch.load(checkpoint_path, map_location=device, weights_only=True)
    except TypeError:
        ckpt = torch.load(checkpoint_path, map_location=device)
    state = ckpt["model_state"] if isinstance(ckpt, dict) and "model_state" in ckpt else ckpt
    model.load_state_dict(state)
    return model


@torch.no_grad()
def predict_scores(model, features: torch.Tensor, batch_size: int = 32, device="cuda", apply_sigmoid: bool = True,
                   swap_tf: bool = True, rank: int = 0, world: int = 1, input_dtype=None) -> torch.Tensor:
    """Scores for a stacked [N,180,321] feature tensor (this rank's shard when world > 1), as one GPU tensor."""
    from .dataloaders import FlatBatcher
    model.eval()
    outs = []
    for feats, _ in FlatBatcher(features, None, batch_size, device=device, rank=rank, world=world, dtype=input_dtype):
        x = feats.transpose(1, 2) if swap_tf else feats          # src/predict.py:104-105
        logits = model(x).squeeze(-1)
        outs.append(torch.sigmoid(logits) if apply_sigmoid else logits)
    return torch.cat(outs) if outs else torch.empty(0, device=device)


@torch.no_grad()
def extract_embeddings(model, features: torch.Tensor, batch_size: int = 256, device="cuda", swap_tf: bool = True,
                       rank: int = 0, world: int = 1, input_dtype=None):
    """Bulk export of the block-3 time-mean embedding [N, 128*F] (what src/embedding_anomaly.py:49-73 collects with
    forward hooks for its OC-SVM / GMM stage) together with the logits [N]: the HIP forward writes the embedding rows
    straight from the block-3 epilogue (`return_embedding=True`), so the export costs one extra 92 KB store per utterance.
    Returns (embeddings, logits) as host tensors (this rank's shard when world > 1)."""
    from .dataloaders import FlatBatcher
    if not hasattr(model, "_eval_forward"):
        raise ValueError("extract_embeddings needs the CNN2D model (the 1D CNN has no [128*F] embedding)")
    model.eval()
    embs, logits = [], []
    for feats, _ in FlatBatcher(features, None, batch_size, device=device, rank=rank, world=world, dtype=input_dtype):
        x = feats.transpose(1, 2) if swap_tf else feats
        lg, e = model(x, return_embedding=True)
        embs.append(e.cpu())
        logits.append(lg.squeeze(-1).cpu())
    if not embs:
        return torch.empty(0, 0), torch.empty(0)
    return torch.cat(embs), torch.cat(logits)


def write_predictions(uttids, scores, out_path: str) -> pd.DataFrame:
    """prediction.pkl exactly as src/predict.py:116-122 writes it (uttid object column, float64 predictions)."""
    scores = [float(s) for s in scores]
    if len(scores) != len(uttids):
        raise ValueError("Number of predictions does not match number of rows in features.pkl")
    df = pd.DataFrame({"uttid": uttids, "predictions": scores})
    df.to_pickle(out_path)
    return df


def build_submission(features_df: pd.DataFrame, prediction_df: pd.DataFrame, student_id: str, first_name: str,
                     last_name: str, nickname: str) -> dict:
    """The submission dict of scripts/generate_submission.py:17-48, with the same checks and error messages:
    {student_id, first_name, last_name, nickname, predictions: DataFrame{uttid, predictions float64}}."""
    import numpy as np
    if len(prediction_df.columns) != 2:
        raise ValueError("prediction.pkl must have exactly 2 columns")
    if "uttid" not in prediction_df.columns or "predictions" not in prediction_df.columns:
        raise ValueError("prediction.pkl must have 'uttid' and 'predictions' columns")
    if "uttid" not in features_df.columns:
        raise ValueError("features.pkl must have 'uttid' column")
    if set(features_df["uttid"].values) != set(prediction_df["uttid"].values):
        raise ValueError("uttid mismatch between features.pkl and prediction.pkl")
    if not all(isinstance(x, (float, np.floating)) for x in prediction_df["predictions"].values):
        prediction_df = prediction_df.copy()
        prediction_df["predictions"] = prediction_df["predictions"].astype(np.float64)
    return {"student_id": student_id, "first_name": first_name, "last_name": last_name, "nickname": nickname,
            "predictions": prediction_df}


def write_submission(features_path: str, prediction_path: str, student_id: str, first_name: str, last_name: str,
                     nickname: str, out_dir: str = ".") -> str:
    """Counterpart of `python scripts/generate_submission.py features.pkl prediction.pkl ID First Last Nick`
    (:38-53): pickles the dict as <ID>-<First>-<Last>-<Nick>.pkl and returns the path."""
    import os
    import pickle
    result = build_submission(pd.read_pickle(features_path), pd.read_pickle(prediction_path), student_id, first_name,
                              last_name, nickname)
    path = os.path.join(out_dir, f"{student_id}-{first_name}-{last_name}-{nickname}.pkl")
    with open(path, "wb") as fh:
        pickle.dump(result, fh)
    return path


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Generate prediction.pkl from a model checkpoint (MI355X HIP path).")
    p.add_argument("--features", required=True, help="Path to features.pkl")
    p.add_argument("--checkpoint", required=True, help="Path to model checkpoint")
    p.add_argument("--model", required=True, choices=["cnn2d", "cnn1d"])
    p.add_argument("--out", required=True, help="Output path for prediction.pkl")
    p.add_argument("--batch-size", type=int, default=32)
    p.add_argument("--num-workers", type=int, default=2)
    p.add_argument("--device", default=None, help="cuda (the HIP path has no CPU fallback)")
    _add_embedding_flag(p)
    p.add_argument("--in-features", type=int, default=180)
    p.add_argument("--dropout", type=float, default=0.3)
    p.add_argument("--apply-sigmoid", action="store_true", default=True)
    p.add_argument("--no-apply-sigmoid", action="store_true", default=False)
    p.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x3"])
    sw = p.add_mutually_exclusive_group()
    sw.add_argument("--swap-tf", dest="swap_tf", action="store_true", help="swap time and feature dims (default)")
    sw.add_argument("--no-swap-tf", dest="swap_tf", action="store_false")
    p.set_defaults(swap_tf=True)
    return p.parse_args(argv)


def _add_embedding_flag(p):
    p.add_argument("--embeddings-out", default=None,
                   help="also write the block-3 embeddings [N, 128*F] (+ uttids, logits) to this .pt file (cnn2d only)")


def main(argv=None):
    from . import distributed as _dist
    _dist.limit_cpu_threads()      # the job's CPU share, not the machine's CPU count (distributed.cpu_budget)
    args = parse_args(argv)
    device = args.device or "cuda"
    apply_sigmoid = False if args.no_apply_sigmoid else args.apply_sigmoid
    model = build_model(args.model, args.in_features, args.dropout, args.precision).to(device)
    load_weights(model, args.checkpoint, device)
    model.eval()
    features_df = pd.read_pickle(args.features)
    if "uttid" not in features_df.columns:
        raise ValueError("features.pkl must contain 'uttid'")
    feats = torch.stack([f.float() for f in features_df["features"]]) if len(features_df) else torch.empty(0, 180, 321)
    scores = predict_scores(model, feats, batch_size=args.batch_size, device=device, apply_sigmoid=apply_sigmoid,
                            swap_tf=args.swap_tf)
    write_predictions(features_df["uttid"].values, scores.cpu().tolist(), args.out)
    if args.embeddings_out:
        emb, logits = extract_embeddings(model, feats, batch_size=max(args.batch_size, 256), device=device, swap_tf=args.swap_tf)
        torch.save({"uttid": list(features_df["uttid"].values), "embeddings": emb, "logits": logits}, args.embeddings_out)


if __name__ == "__main__":
    main()
