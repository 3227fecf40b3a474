"""Scoring and evaluation -- host-side counterpart of the reference's src/evaluation.py (and the official scorer
scripts/evaluation.py).  EER is O(N log N) on N <= a few thousand scores and stays on the host exactly as in the
reference; the model forward inside `evaluate` is the HIP path.

Signatures kept: calculate_eer(scores, labels) -> (eer, threshold); evaluate(model, dataloader, criterion=None,
device=..., apply_sigmoid=False, swap_tf=False) -> (metrics, scores, labels); verify_uttid_alignment(features_path,
labels_path); confusion_at_threshold(scores, labels, threshold).
"""
from __future__ import annotations

import argparse

import numpy as np
import pandas as pd
import torch

_THRESHOLD_EPS = 1e-6


def calculate_eer(scores, labels):
    """Equal error rate of `scores` (higher = bonafide) against 0/1 `labels` (1 = bonafide).

    Restates scripts/evaluation.py:7-39 (== src/evaluation.py:12-48): sweep the threshold over the ascending-sorted
    scores; FAR_k = spoof samples still accepted after rejecting the k lowest, FRR_k = bonafide samples among those k;
    the EER is the mean of the two rates where they are closest.  np.argsort's default (unstable) ordering is kept on
    purpose: ties must break the way the reference's scorer breaks them.
    """
    s = np.array(scores)
    y = np.array(labels)
    n_pos = np.sum(y)
    n_neg = len(y) - n_pos
    if n_pos == 0 or n_neg == 0:
        return 0.0, 0.0
    order = np.argsort(s)
    s_sorted = s[order]
    y_sorted = y[order]
    far = np.concatenate([[1.0], (n_neg - np.cumsum(y_sorted == 0)) / n_neg])
    frr = np.concatenate([[0.0], np.cumsum(y_sorted == 1) / n_pos])
    k = np.argmin(np.abs(far - frr))
    eer = (far[k] + frr[k]) / 2.0
    if k == 0:
        thr = s_sorted[0] - _THRESHOLD_EPS
    elif k == len(s_sorted):
        thr = s_sorted[-1] + _THRESHOLD_EPS
    else:
        thr = s_sorted[k - 1]
    return float(eer), float(thr)


def confusion_at_threshold(scores, labels, threshold):
    """(tp, fp, tn, fn, far, frr) with prediction = score > threshold (scripts/evaluation.py:42-56)."""
    s = np.array(scores)
    y = np.array(labels).astype(int)
    hit = s > threshold
    tp = int(np.sum(hit & (y == 1)))
    fn = int(np.sum(~hit & (y == 1)))
    fp = int(np.sum(hit & (y == 0)))
    tn = int(np.sum(~hit & (y == 0)))
    far = fp / (fp + tn) if (fp + tn) > 0 else 0.0
    frr = fn / (tp + fn) if (tp + fn) > 0 else 0.0
    return tp, fp, tn, fn, float(far), float(frr)


def evaluate(model, dataloader, criterion=None, device="cuda", apply_sigmoid=False, swap_tf: bool = False):
    """Run `model` over a labelled loader; returns (metrics{avg_loss, eer, threshold}, scores, labels).

    Counterpart of src/evaluation.py:51-104.  Differences that do not change results: logits stay on the GPU until
    the end of the loop (one device->host copy instead of one `.item()`/`.tolist()` sync per batch).
    """
    model.eval()
    logit_chunks, label_chunks = [], []
    loss_sum = None
    count = 0
    with torch.no_grad():
        for features, batch_labels in dataloader:
            features = features.to(device, non_blocking=True)
            batch_labels = batch_labels.to(device, non_blocking=True)
            if swap_tf:
                features = features.transpose(1, 2)
            logits = model(features).squeeze(-1)
            if criterion is not None:
                term = criterion(logits, batch_labels).detach().double() * batch_labels.size(0)
                loss_sum = term if loss_sum is None else loss_sum + term
                count += batch_labels.size(0)
            logit_chunks.append(logits.detach())
            label_chunks.append(batch_labels.detach())
    scores, labels = [], []
    if logit_chunks:
        all_logits = torch.cat(logit_chunks)
        out = torch.sigmoid(all_logits) if apply_sigmoid else all_logits
        scores = out.cpu().tolist()
        labels = torch.cat(label_chunks).cpu().tolist()
    avg_loss = float(loss_sum.item() / count) if count > 0 else None
    eer, threshold = (None, None)
    if scores and labels:
        eer, threshold = calculate_eer(scores, labels)
    return {"avg_loss": avg_loss, "eer": eer, "threshold": threshold}, scores, labels


def evaluate_sharded(model, features: torch.Tensor, labels: torch.Tensor, criterion=None, device="cuda",
                     apply_sigmoid: bool = False, swap_tf: bool = True, batch_size: int = 32, rank: int = 0,
                     world: int = 1, input_dtype=None):
    """`evaluate` for data-parallel runs ("dev evaluation inside training = sharded inference +
    gather"): rank r scores the contiguous utterance range [r*ceil(N/world), ...) of the stacked set, the per-rank
    logit vectors are gathered in rank order, and every rank computes the SAME loss / EER from the same gathered
    vector -- so best-checkpoint and early-stop decisions cannot diverge between ranks.  No data-path collective."""
    from . import distributed as dfa_dist
    from .dataloaders import FlatBatcher
    model.eval()
    chunks = []
    with torch.no_grad():
        for feats, _ in FlatBatcher(features, None, batch_size, device=device, rank=rank, world=world, dtype=input_dtype):
            x = feats.transpose(1, 2) if swap_tf else feats
            chunks.append(model(x).squeeze(-1).detach())
    local = torch.cat(chunks).double().cpu().numpy() if chunks else np.zeros(0)
    logits = dfa_dist.gather_scores(local, device=torch.device(device) if str(device).startswith("cuda") else None)
    y = labels.double().cpu().numpy()
    if len(logits) != len(y):
        raise ValueError(f"gathered {len(logits)} scores for {len(y)} labels")
    avg_loss = None
    if criterion is not None and len(y):
        avg_loss = float(criterion(torch.from_numpy(logits).float(), torch.from_numpy(y).float()).item())
    scores = (1.0 / (1.0 + np.exp(-logits))) if apply_sigmoid else logits
    eer, threshold = calculate_eer(scores.tolist(), y.tolist()) if len(y) else (None, None)
    return {"avg_loss": avg_loss, "eer": eer, "threshold": threshold}, scores.tolist(), y.tolist()


def verify_uttid_alignment(features_path: str, labels_path: str) -> None:
    """ValueError unless both pickles carry 'uttid' and describe the same utterances (src/evaluation.py:107-124)."""
    feats = pd.read_pickle(features_path)
    labs = pd.read_pickle(labels_path)
    if "uttid" not in feats.columns:
        raise ValueError("features.pkl must contain 'uttid'")
    if "uttid" not in labs.columns:
        raise ValueError("labels.pkl must contain 'uttid'")
    both = pd.merge(feats[["uttid"]], labs[["uttid"]], on="uttid", how="inner")
    if len(both) != len(feats) or len(both) != len(labs):
        raise ValueError("uttid mismatch between features and labels")


def score_prediction_file(prediction_path: str, labels_path: str):
    """What `python scripts/evaluation.py prediction.pkl labels.pkl` computes (scripts/evaluation.py:59-90):
    returns dict(eer, threshold, tp, fp, tn, fn, far, frr)."""
    pred = pd.read_pickle(prediction_path)
    labs = pd.read_pickle(labels_path)
    if "uttid" not in pred.columns or "predictions" not in pred.columns:
        raise ValueError("prediction.pkl must have 'uttid' and 'predictions' columns")
    if "uttid" not in labs.columns or "label" not in labs.columns:
        raise ValueError("labels.pkl must have 'uttid' and 'label' columns")
    merged = pd.merge(pred, labs, on="uttid", how="inner")
    if len(merged) != len(pred) or len(merged) != len(labs):
        raise ValueError("uttid mismatch between prediction and labels")
    s, y = merged["predictions"].values, merged["label"].values
    eer, thr = calculate_eer(s, y)
    tp, fp, tn, fn, far, frr = confusion_at_threshold(s, y, thr)
    return {"eer": eer, "threshold": thr, "tp": tp, "fp": fp, "tn": tn, "fn": fn, "far": far, "frr": frr}


def _main():
    p = argparse.ArgumentParser(description="Evaluate a model checkpoint on a labeled dataset (MI355X HIP path).")
    p.add_argument("--features", required=True, help="Path to features.pkl")
    p.add_argument("--labels", required=True, help="Path to labels.pkl")
    p.add_argument("--checkpoint", required=True)
    p.add_argument("--model", required=True, choices=["cnn2d", "cnn1d"])
    p.add_argument("--batch-size", type=int, default=32)
    p.add_argument("--num-workers", type=int, default=2)
    p.add_argument("--device", default="cuda")
    p.add_argument("--in-features", type=int, default=180)
    p.add_argument("--dropout", type=float, default=0.2)
    p.add_argument("--apply-sigmoid", action="store_true")
    p.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x3"])
    sw = p.add_mutually_exclusive_group()
    sw.add_argument("--swap-tf", dest="swap_tf", action="store_true")
    sw.add_argument("--no-swap-tf", dest="swap_tf", action="store_false")
    p.set_defaults(swap_tf=True)
    args = p.parse_args()

    from .dataloaders import make_loader
    from .predict import build_model, load_weights
    verify_uttid_alignment(args.features, args.labels)
    model = build_model(args.model, args.in_features, args.dropout, args.precision).to(args.device)
    load_weights(model, args.checkpoint, args.device)
    loader = make_loader(args.features, args.labels, batch_size=args.batch_size, num_workers=args.num_workers)
    metrics, _, _ = evaluate(model, loader, criterion=torch.nn.BCEWithLogitsLoss(), device=args.device,
                             apply_sigmoid=args.apply_sigmoid, swap_tf=args.swap_tf)
    print(f"loss={metrics['avg_loss']:.6f}  EER={metrics['eer']:.6f}  threshold={metrics['threshold']:.6f}")


if __name__ == "__main__":
    _main()
