"""Checkpoint files in the reference's format (src/training/checkpoint.py:8-109): a torch.save'd dict
{"model_state", "optimizer_state", "epoch", "config"[, "scheduler_state"]}.  Because the dfa_amd models keep the
reference's state_dict keys, checkpoints are interchangeable in both directions."""
from __future__ import annotations

from pathlib import Path
from typing import Any, Dict, Optional

import torch

_CONFIG_KEYS = ("batch_size", "num_workers", "lr", "weight_decay", "lr_scheduler", "lr_scheduler_metric",
                "lr_scheduler_factor", "lr_scheduler_patience", "lr_scheduler_threshold", "lr_scheduler_min_lr",
                "in_features", "hidden_dim", "dropout", "dropout_mlp", "dropout_cnn", "pool_bins")


def build_config_dict(args) -> Dict[str, Any]:
    name = getattr(args, "model", None)
    if name is None:
        name = getattr(args, "model_name", None)
    cfg = {"model_name": name}
    cfg.update({k: getattr(args, k, None) for k in _CONFIG_KEYS})
    return cfg


def save_checkpoint(model, optimizer, epoch: int, args, path: str, scheduler: Optional[Any] = None) -> None:
    blob = {"model_state": model.state_dict(), "optimizer_state": optimizer.state_dict(), "epoch": epoch,
            "config": build_config_dict(args)}
    if scheduler is not None:
        blob["scheduler_state"] = scheduler.state_dict()
    Path(path).parent.mkdir(parents=True, exist_ok=True)
    torch.save(blob, path)


def load_checkpoint(path: str, model=None, optimizer=None, device: str = "cpu", scheduler=None) -> Dict[str, Any]:
    if not Path(path).exists():
        raise FileNotFoundError(f"Checkpoint not found: {path}")
    blob = torch.load(path, map_location=device)
    if model is not None and "model_state" in blob:
        model.load_state_dict(blob["model_state"])
    if optimizer is not None and "optimizer_state" in blob:
        optimizer.load_state_dict(blob["optimizer_state"])
    if scheduler is not None and "scheduler_state" in blob:
        scheduler.load_state_dict(blob["scheduler_state"])
    return blob
