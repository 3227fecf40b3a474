"""Training driver -- counterpart of src/train.py (same `train_one_epoch` signature and the CLI flags that concern the
CNN2D/CNN1D path), with the model step on the MI355X HIP path.

Two step engines:
  * reference-style (default for `train_one_epoch`): `loss = criterion(model(x)); loss.backward(); optimizer.step()`
    with any torch criterion / optimizer -- the model's train-mode forward/backward are C-ABI calls behind a
    torch.autograd.Function;
  * `--native`: the whole step (forward, BCE + label smoothing, backward, one flat gradient all-reduce, fused AdamW)
    on the C ABI via `NativeTrainer`; one process per GPU under torchrun gives data-parallel training.
"""
from __future__ import annotations

import argparse
import os
import random
from typing import Callable, Optional

import numpy as np
import torch
from torch import nn

from . import distributed as dfa_dist
from .augmentation import FusedAugment, channel_drop, compose, gaussian_jitter, spec_augment, time_shift
from .dataloaders import FlatBatcher, IndexedFlatBatcher, ResidentBatcher, make_loader, open_flat, train_shard_indices
from .dataset import AudioDeepfakeDataset
from .evaluation import evaluate, evaluate_sharded
from .model import CNN2D
from .training import save_checkpoint


def train_one_epoch(model, dataloader, criterion, optimizer, device: str = "cuda", batch_context=None,
                    augment_fn: Optional[Callable[[torch.Tensor], torch.Tensor]] = None, swap_tf: bool = False):
    """One pass over `dataloader` (src/train.py:31-91); returns the sample-weighted mean training loss.
    The running loss is accumulated on the device; it is synchronised to the host only when a `batch_context`
    (progress display) asks for it, not once per batch."""
    model.train()
    loss_sum = None
    count = 0
    for batch_idx, (features, labels) in enumerate(dataloader):
        features = features.to(device, non_blocking=True)
        labels = labels.to(device, non_blocking=True)
        if swap_tf:
            features = features.transpose(1, 2)
        if augment_fn is not None:
            features = augment_fn(features)
        logits = model(features).squeeze(-1)
        loss = criterion(logits, labels)
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        term = loss.detach().double() * labels.size(0)
        loss_sum = term if loss_sum is None else loss_sum + term
        count += labels.size(0)
        if batch_context is not None and count > 0:
            batch_context.update_batch(_BatchMetrics(batch_idx, float(loss_sum.item()) / count, labels.size(0)))
    return (float(loss_sum.item()) / count) if count > 0 else None


class _BatchMetrics:
    def __init__(self, batch_idx, running_loss, batch_size):
        self.batch_idx, self.running_loss, self.batch_size = batch_idx, running_loss, batch_size


def train_one_epoch_native(trainer, batcher, augment_fn=None, swap_tf: bool = True):
    """One pass with the all-native step; `batcher` yields (stored-layout features [b,180,321], labels [b]) on the GPU."""
    total, count = None, 0
    # bf16 storage mode of the CNN2D without augmentation: the fp32 batch is rounded to bf16 on the compute stream (the same
    # round-to-nearest-even the first kernel applies on load), so block 1 takes its matrix-core passes, which read bf16 features:
    # 47.3 k against 42.2 k utterances/s for the whole epoch loop (tools/gpu_train_epoch_probe.py; the cast on the copy stream,
    # behind the H2D, measured 32 k)
    to_bf16 = (augment_fn is None and getattr(trainer, "kind", "") == "cnn2d" and getattr(trainer.model, "precision", "") == "bf16")
    for feats, labels in batcher:
        if to_bf16 and feats.dtype == torch.float32:
            feats = feats.to(torch.bfloat16)
        x = feats.transpose(1, 2) if swap_tf else feats
        if augment_fn is not None:
            x = augment_fn(x)
        loss = trainer.step(x, labels)
        term = loss.detach().double().squeeze() * labels.size(0)
        total = term if total is None else total + term
        count += labels.size(0)
    return (float(total.item()) / count) if count else None


def build_augment_fn(args, fused: bool = False, fold: bool = False):
    """fused = True (GPU training): the whole pipeline as one HIP pass (augmentation.FusedAugment), same parameter draws;
    fold = "cnn2d" / "cnn1d" (True = "cnn2d"): not even a pass -- the parameters are armed on the context and that model's
    training kernels that read x apply them in their loads."""
    if fused and (args.spec_augment or args.time_shift or args.channel_drop or args.gaussian_jitter):
        return FusedAugment(fold=fold, spec_augment=args.spec_augment, time_mask_ratio=args.time_mask_ratio,
                            feature_mask=args.feature_mask, feature_mask_ratio=args.feature_mask_ratio,
                            time_shift=args.time_shift, time_shift_ratio=args.time_shift_ratio,
                            channel_drop=args.channel_drop, channel_drop_prob=args.channel_drop_prob,
                            gaussian_jitter=args.gaussian_jitter, gaussian_jitter_std=args.gaussian_jitter_std)
    fns = []
    if args.spec_augment:
        fns.append(lambda x: spec_augment(x, time_mask_ratio=args.time_mask_ratio,
                                          feature_mask_ratio=args.feature_mask_ratio, apply_time_mask=True,
                                          apply_feature_mask=args.feature_mask))
    if args.time_shift:
        fns.append(lambda x: time_shift(x, max_shift_ratio=args.time_shift_ratio))
    if args.channel_drop:
        fns.append(lambda x: channel_drop(x, drop_prob=args.channel_drop_prob))
    if args.gaussian_jitter:
        fns.append(lambda x: gaussian_jitter(x, std=args.gaussian_jitter_std))
    return compose(*fns) if fns else None


def make_criterion(label_smoothing: float):
    if not (0.0 <= label_smoothing < 0.5):
        raise ValueError("--label-smoothing must be in [0, 0.5)")
    bce = nn.BCEWithLogitsLoss()
    eps = float(label_smoothing)

    def criterion(logits, y):
        return bce(logits, y * (1.0 - eps) + 0.5 * eps if eps > 0 else y)
    return criterion


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Train CNN2D for audio deepfake detection on MI355X.")
    p.add_argument("--train-features", default="data/train/features.pkl")
    p.add_argument("--train-labels", default="data/train/labels.pkl")
    p.add_argument("--dev-features", default="data/dev/features.pkl")
    p.add_argument("--dev-labels", default="data/dev/labels.pkl")
    p.add_argument("--model", default="cnn2d", choices=["cnn2d", "cnn1d"])
    p.add_argument("--batch-size", type=int, default=32)
    p.add_argument("--num-workers", type=int, default=2)
    p.add_argument("--epochs", type=int, default=10)
    p.add_argument("--lr", type=float, default=1e-3)
    p.add_argument("--weight-decay", type=float, default=0.0)
    p.add_argument("--early-stop", type=int, default=0)
    p.add_argument("--lr-scheduler", default="none", choices=["none", "plateau"])
    p.add_argument("--lr-scheduler-metric", default="dev_eer", choices=["dev_eer", "dev_loss"])
    p.add_argument("--lr-scheduler-factor", type=float, default=0.5)
    p.add_argument("--lr-scheduler-patience", type=int, default=2)
    p.add_argument("--lr-scheduler-threshold", type=float, default=1e-4)
    p.add_argument("--lr-scheduler-min-lr", type=float, default=1e-6)
    p.add_argument("--device", default="cuda")
    p.add_argument("--in-features", type=int, default=180)
    p.add_argument("--dropout", type=float, default=0.2)
    p.add_argument("--checkpoint-dir", default="checkpoints")
    p.add_argument("--run-name", default="")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--spec-augment", action="store_true")
    p.add_argument("--time-mask-ratio", type=float, default=0.2)
    p.add_argument("--feature-mask-ratio", type=float, default=0.1)
    p.add_argument("--feature-mask", action="store_true")
    p.add_argument("--time-shift", action="store_true")
    p.add_argument("--time-shift-ratio", type=float, default=0.1)
    p.add_argument("--channel-drop", action="store_true")
    p.add_argument("--channel-drop-prob", type=float, default=0.1)
    p.add_argument("--gaussian-jitter", action="store_true")
    p.add_argument("--gaussian-jitter-std", type=float, default=0.01)
    p.add_argument("--label-smoothing", type=float, default=0.0)
    p.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    p.add_argument("--native", action="store_true", help="all-native step (fused AdamW, flat-gradient all-reduce)")
    p.add_argument("--resident", default="auto", choices=["auto", "on", "off"],
                   help="flat-buffer trainers: keep the whole training set in GPU memory (uploaded once; batches are device-side row "
                        "gathers) -- auto: when it needs at most half of the free memory")
    p.add_argument("--sync-bn", action="store_true",
                   help="data-parallel training: BatchNorm statistics over the global batch (N ranks x B train like one rank x N*B); "
                        "default: each rank's own statistics, as torch DistributedDataParallel")
    sw = p.add_mutually_exclusive_group()
    sw.add_argument("--swap-tf", dest="swap_tf", action="store_true")
    sw.add_argument("--no-swap-tf", dest="swap_tf", action="store_false")
    p.set_defaults(swap_tf=True)
    return p.parse_args(argv)


def set_seed(seed: int) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def main(argv=None):
    from . import distributed as _dist
    _dist.limit_cpu_threads()      # the job's CPU share, not the machine's CPU count (distributed.cpu_budget)
    args = parse_args(argv)
    set_seed(args.seed)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", local_rank) if args.device.startswith("cuda") else torch.device(args.device)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    rank, world = dfa_dist.init(device=device)
    out_dir = os.path.join(args.checkpoint_dir, args.run_name) if args.run_name else args.checkpoint_dir
    best_path, last_path = os.path.join(out_dir, f"{args.model}_best.pt"), os.path.join(out_dir, f"{args.model}_last.pt")

    if args.model == "cnn1d":
        from .model_cnn1d import CNN1D
        model = CNN1D(in_features=args.in_features, dropout=args.dropout).to(device)
    else:
        model = CNN2D(in_features=args.in_features, dropout=args.dropout, precision=args.precision).to(device)
    # dropout masks (and the jitter noise below) are Philox streams keyed by the run seed AND the rank: the ranks of one
    # data-parallel step must not draw identical masks
    model._drop_seed = dfa_dist.rank_seed(args.seed if args.seed else torch.initial_seed(), rank)
    weight_decay = args.weight_decay if args.weight_decay > 0 else 0.01      # AdamW default of src/train.py:321-325
    criterion = make_criterion(args.label_smoothing)
    # batches are on the GPU when it is applied; for both classifiers the augmentation is folded into the kernels' loads
    augment_fn = build_augment_fn(args, fused=(device.type == "cuda"), fold=(args.model if device.type == "cuda" and args.model in ("cnn2d", "cnn1d") else False))
    if isinstance(augment_fn, FusedAugment):
        augment_fn.seed = dfa_dist.rank_seed(augment_fn.seed, rank)

    flat_mode = args.native or world > 1          # one flat parameter / gradient buffer, one all-reduce per step
    trainer = None
    if flat_mode:
        from .training.train_step import FlatTrainer, NativeTrainer
        # all-C-ABI step for both classifiers: 464,644-byte (CNN2D) / 195,204-byte (CNN1D) flat gradient, one all-reduce
        trainer = NativeTrainer(model, lr=args.lr, weight_decay=weight_decay, label_smoothing=args.label_smoothing,
                                sync_bn=bool(getattr(args, "sync_bn", False)) and world > 1)
        dfa_dist.broadcast_parameters_(trainer.flat_p)
        dfa_dist.average_tensors_(dfa_dist.bn_running_stats(model))
        # Flat, memory-mapped sources: a rank reads only the rows it consumes (1 / world of the training set per epoch, its
        # contiguous share of the dev set); a features.pkl is converted ONCE, by rank 0 (dataloaders.open_flat)
        cache = os.path.join(out_dir, "flat_cache")
        feats, labels, _ = open_flat(args.train_features, args.train_labels, cache, rank, world, "train")
        dev_feats, dev_labels, _ = open_flat(args.dev_features, args.dev_labels, cache, rank, world, "dev")
        optimizer = trainer
    else:
        optimizer = torch.optim.AdamW(model.parameters(), lr=args.lr, weight_decay=weight_decay)
        train_loader = make_loader(args.train_features, args.train_labels, batch_size=args.batch_size,
                                   num_workers=args.num_workers, shuffle=True)
        dev_loader = make_loader(args.dev_features, args.dev_labels, batch_size=args.batch_size,
                                 num_workers=args.num_workers, shuffle=False)
    scheduler = None
    if args.lr_scheduler == "plateau":            # src/train.py:332-341
        kw = dict(mode="min", factor=args.lr_scheduler_factor, patience=args.lr_scheduler_patience,
                  threshold=args.lr_scheduler_threshold, min_lr=args.lr_scheduler_min_lr)
        scheduler = trainer.plateau_scheduler(**kw) if trainer is not None else \
            torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, **kw)

    best_eer = best_train = best_dev = None
    no_improve, last_epoch = 0, 0
    resident = None
    for epoch in range(1, args.epochs + 1):
        if flat_mode:
            # every rank draws the SAME permutation and takes its rows of every global batch: equal step counts and equal
            # local batch sizes on all ranks (dataloaders.train_shard_indices)
            perm = torch.randperm(feats.shape[0], generator=torch.Generator().manual_seed(args.seed + epoch))
            idx = train_shard_indices(perm, args.batch_size, rank, world)
            if resident is None and args.resident != "off" and device.type == "cuda":
                # MI355X-first: the whole training set lives in HBM (288 GB) when it fits -- uploaded once, batches are device-side
                # row gathers; "auto" takes it when the set needs at most half of the free memory
                store = torch.bfloat16 if (args.precision == "bf16" and args.model == "cnn2d" and augment_fn is None) else None
                if args.resident == "on" or ResidentBatcher.fits(feats, device, store):
                    resident = ResidentBatcher(feats, labels, args.batch_size, device=device, dtype=store)
                else:
                    resident = False
            batcher = resident.epoch(idx) if resident else IndexedFlatBatcher(feats, labels, idx, args.batch_size, device=device)
            if isinstance(trainer, FlatTrainer):
                train_loss = train_one_epoch(model, batcher, criterion, trainer, device=device, augment_fn=augment_fn,
                                             swap_tf=args.swap_tf)
            else:
                train_loss = train_one_epoch_native(trainer, batcher, augment_fn, args.swap_tf)
            train_loss = dfa_dist.mean_scalar(train_loss, device)
            # BatchNorm running statistics come from rank-local batches: average them so that every rank evaluates (and
            # rank 0 checkpoints) the same model
            dfa_dist.average_tensors_(dfa_dist.bn_running_stats(model))
            model._prepared = None
            metrics, _, _ = evaluate_sharded(model, dev_feats, dev_labels, criterion=criterion, device=device,
                                             swap_tf=args.swap_tf, batch_size=args.batch_size, rank=rank, world=world)
        else:
            train_loss = train_one_epoch(model, train_loader, criterion, optimizer, device=device,
                                         augment_fn=augment_fn, swap_tf=args.swap_tf)
            metrics, _, _ = evaluate(model, dev_loader, criterion=criterion, device=device, swap_tf=args.swap_tf)
        eer, dev_loss = metrics["eer"], metrics["avg_loss"]
        is_best = False
        if eer is not None:
            if best_eer is None or eer < best_eer:
                is_best, best_eer, best_train, best_dev, no_improve = True, eer, train_loss, dev_loss, 0
            else:
                no_improve += 1
                if (abs(eer - best_eer) <= 1e-4 and None not in (train_loss, dev_loss, best_train, best_dev)
                        and train_loss < best_train - 1e-6 and dev_loss < best_dev - 1e-6):
                    is_best, best_train, best_dev = True, train_loss, dev_loss
        if scheduler is not None:
            metric = dev_loss if args.lr_scheduler_metric == "dev_loss" else eer
            if metric is not None:
                scheduler.step(metric)             # the metric is identical on every rank (gathered scores)
        if rank == 0:
            print(f"epoch {epoch}: train_loss={train_loss:.6f} dev_loss={dev_loss:.6f} dev_eer={eer:.6f}"
                  + ("  *best*" if is_best else ""))
            if is_best:
                save_checkpoint(model, optimizer, epoch, args, best_path, scheduler=scheduler)
        last_epoch = epoch
        if args.early_stop and no_improve >= args.early_stop:
            break                                  # same decision on every rank: it depends on gathered metrics only
    if rank == 0:
        save_checkpoint(model, optimizer, last_epoch, args, last_path, scheduler=scheduler)


if __name__ == "__main__":
    main()
