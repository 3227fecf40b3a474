"""Mean-of-sigmoids ensemble over checkpoints -- counterpart of src/ensemble.py (same flags: --checkpoints arch:path ...).

    python -m dfa_amd.ensemble --checkpoints cnn2d:a/cnn2d_best.pt cnn1d:b/cnn1d_best.pt --dev-features F --dev-labels L

Every member scores the labelled set on the HIP path (utterance-sharded over ranks, no data-path collective, the score
vectors gathered with one tensor all-gather); the per-model EERs, the mean of the sigmoid scores (src/ensemble.py:121) and
its EER are computed on the host exactly as the reference does."""
from __future__ import annotations

import argparse

import numpy as np
import pandas as pd
import torch

from . import distributed as dfa_dist
from . import fusion
from .evaluation import calculate_eer
from .hybrid_ensemble import _stack, score_models
from .model import CNN2D
from .model_cnn1d import CNN1D
from .predict import load_weights


def load_model(arch: str, checkpoint_path: str, device, in_features: int = 180, dropout: float = 0.2, precision: str = "fp32"):
    """src/ensemble.py:32-49: instantiate, load (wrapped or bare state_dict), eval mode."""
    if arch == "cnn1d":
        model = CNN1D(in_features=in_features, dropout=dropout)
    elif arch == "cnn2d":
        model = CNN2D(in_features=in_features, dropout=dropout, precision=precision)
    else:
        raise ValueError(f"unknown architecture '{arch}' (cnn2d or cnn1d)")
    return load_weights(model.to(device), checkpoint_path, device).eval()


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Ensemble-average sigmoid scores from multiple checkpoints (MI355X HIP path).")
    p.add_argument("--checkpoints", nargs="+", required=True, help="arch:path pairs, e.g. cnn2d:checkpoints/run/cnn2d_best.pt")
    p.add_argument("--dev-features", default="data/dev/features.pkl")
    p.add_argument("--dev-labels", default="data/dev/labels.pkl")
    p.add_argument("--batch-size", type=int, default=32)
    p.add_argument("--device", default=None)
    p.add_argument("--in-features", type=int, default=180)
    p.add_argument("--dropout", type=float, default=0.2)
    p.add_argument("--swap-tf", action="store_true", default=True)
    p.add_argument("--no-swap-tf", dest="swap_tf", action="store_false")
    p.add_argument("--precision", default="fp32", choices=["fp32", "bf16", "bf16x3"])
    p.add_argument("--out", default=None, help="optional: write the ensemble scores as a prediction.pkl")
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if not args.swap_tf:
        raise ValueError("--no-swap-tf: the stored layout is [180, 321]; the HIP models read it through the transposed view")
    device = args.device or "cuda"
    rank, world = dfa_dist.init()
    table = pd.merge(pd.read_pickle(args.dev_features), pd.read_pickle(args.dev_labels), on="uttid", how="inner") \
        .reset_index(drop=True)                                                      # src/dataset.py:28 (make_loader's merge)
    labels = [float(v) for v in table["label"].values]
    stored = _stack(table)
    dev = torch.device(device) if str(device).startswith("cuda") else None
    all_scores, report = [], []
    for spec in args.checkpoints:
        arch, path = spec.split(":", 1)
        model = load_model(arch, path, device, args.in_features, args.dropout, args.precision)
        local = score_models(stored, cnn2d=model if arch == "cnn2d" else None, cnn1d=model if arch == "cnn1d" else None,
                             batch_size=args.batch_size, device=device, rank=rank, world=world)[arch]
        scores = dfa_dist.gather_scores(local, device=dev)
        all_scores.append(scores)
        eer, thr = calculate_eer(scores.tolist(), labels)
        report.append((arch, path, eer, thr))
        if rank == 0:
            print(f"  {arch:6s}  {path}\n         EER={eer:.6f}  threshold={thr:.6f}")
    ens = fusion.ensemble_mean(all_scores)                                            # src/ensemble.py:121
    eer, thr = calculate_eer(ens.tolist(), labels)
    if rank == 0:
        print(f"\n{'=' * 60}\nEnsemble of {len(all_scores)} models\n  EER      = {eer:.6f}\n  threshold= {thr:.6f}\n{'=' * 60}")
        if args.out:
            pd.DataFrame({"uttid": table["uttid"].values, "predictions": ens.astype(np.float64)}).to_pickle(args.out)
    return {"members": report, "eer": eer, "threshold": thr, "scores": ens}


if __name__ == "__main__":
    main()
