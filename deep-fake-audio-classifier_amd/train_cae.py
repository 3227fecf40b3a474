"""Auto-encoder trainer -- counterpart of src/train_cae.py (same CLI flags and artefacts: cae_best.pt / cae_last.pt in
the reference checkpoint format plus normalizer.pt), with the model step on the MI355X HIP path.

Bonafide (label == 1) utterances only; inputs are z-scored per feature dim with statistics fitted on the bonafide
training set; loss = MSELoss(reconstruction, input); AdamW(lr 1e-4, wd 1e-4); ReduceLROnPlateau on the validation MSE;
best checkpoint = lowest validation MSE; early stopping on patience."""
from __future__ import annotations

import argparse
import os
import random

import numpy as np
import torch
from torch import nn
from torch.utils.data import DataLoader

from . import distributed as dfa_dist
from .dataloaders import train_shard_indices
from .dataset_cae import BonafideDataset, FeatureNormalizer, build_normalizer, fit_normalizer_sharded
from .model_cae import ConvAutoencoder
from .training import save_checkpoint


def train_one_epoch(model, dataloader, criterion, optimizer, device="cuda"):
    """Mean reconstruction loss over the epoch (src/train_cae.py:58-82); the loss stays on the device until the end."""
    model.train()
    total, count = None, 0
    native = hasattr(optimizer, "flat_g") and hasattr(optimizer, "step") and type(optimizer).__name__ == "CaeNativeTrainer"
    for x in dataloader:
        x = x.to(device, non_blocking=True)
        if native:                       # the whole step on the C ABI (criterion is MSELoss(recon, x) by construction)
            loss = optimizer.step(x)
        else:
            recon, _ = model(x)
            loss = criterion(recon, x)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
        term = loss.detach().double() * x.size(0)
        total = term if total is None else total + term
        count += x.size(0)
    return (float(total.item()) / count) if count else None


@torch.no_grad()
def validate_reconstruction(model, dataloader, device="cuda"):
    """Mean per-sample MSE on a (bonafide) validation loader (src/train_cae.py:85-105), via the fused score kernel."""
    model.eval()
    total, count = None, 0
    for x in dataloader:
        mse = model.score(x.to(device, non_blocking=True)).double().sum()
        total = mse if total is None else total + mse
        count += x.size(0)
    return (float(total.item()) / count) if count else None


@torch.no_grad()
def validate_reconstruction_sharded(model, dataset, batch_size, num_workers, device, rank, world):
    """validate_reconstruction over this rank's contiguous shard of `dataset`, then ONE all-reduce of (sum, count): every
    rank returns the same validation MSE, so scheduler / best / early-stop decisions agree across ranks."""
    import torch.distributed as dist
    from torch.utils.data import Subset
    lo, hi = dfa_dist.shard_range(len(dataset), rank, world)
    loader = DataLoader(Subset(dataset, range(lo, hi)), batch_size=batch_size, shuffle=False, num_workers=num_workers)
    model.eval()
    total = torch.zeros(2, dtype=torch.float64, device=device)
    for x in loader:
        total[0] += model.score(x.to(device, non_blocking=True)).double().sum()
        total[1] += x.size(0)
    if world > 1:
        t = total if dist.get_backend() == "nccl" else total.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total = t
    return float(total[0].item() / total[1].item()) if float(total[1].item()) > 0 else None


class _NormalizedBatches:
    """This rank's batches of an epoch from the flat source: IndexedFlatBatcher rows [b, F, T] (pinned staging, async H2D) ->
    the [b, T, F] view the model takes -> z-score on the device (src/dataset_cae.py:27-30, the dataset's transform)."""

    def __init__(self, batcher, mean, std, dtype=None):
        self.batcher, self.mean, self.std, self.dtype = batcher, mean, std, dtype

    def __len__(self):
        return len(self.batcher)

    def __iter__(self):
        for f, _ in self.batcher:
            x = (f.transpose(1, 2).float() - self.mean) / self.std
            # bf16 storage mode: the z-scored batch is held in bf16 (what the first kernel would round it to on load), so block 1 takes
            # its matrix-core passes, which read bf16 features
            yield x if self.dtype is None else x.to(self.dtype)


@torch.no_grad()
def validate_reconstruction_flat(model, feats, rows, mean, std, batch_size, device, rank, world):
    """validate_reconstruction over this rank's contiguous share of the bonafide dev rows of the flat source, the z-score fused
    into the score kernel's loads (model.score(x, mean, std)); ONE all-reduce of (sum, count)."""
    import torch.distributed as dist
    from .dataloaders import IndexedFlatBatcher
    lo, hi = dfa_dist.shard_range(rows.numel(), rank, world)
    model.eval()
    total = torch.zeros(2, dtype=torch.float64, device=device)
    if hi > lo:
        for f, _ in IndexedFlatBatcher(feats, None, rows[lo:hi], batch_size, device=device):
            total[0] += model.score(f.transpose(1, 2), mean, std).double().sum()
            total[1] += f.shape[0]
    if world > 1:
        t = total if dist.get_backend() == "nccl" else total.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total = t
    return float(total[0].item() / total[1].item()) if float(total[1].item()) > 0 else None


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Train the convolutional auto-encoder for anomaly detection (MI355X).")
    p.add_argument("--train-features", default="data/train/features.pkl")
    p.add_argument("--train-labels", default="data/train/labels.pkl")
    p.add_argument("--dev-features", default="data/dev/features.pkl")
    p.add_argument("--dev-labels", default="data/dev/labels.pkl")
    p.add_argument("--batch-size", type=int, default=32)
    p.add_argument("--num-workers", type=int, default=2)
    p.add_argument("--epochs", type=int, default=80)
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--weight-decay", type=float, default=1e-4)
    p.add_argument("--early-stop", type=int, default=10)
    p.add_argument("--lr-scheduler-patience", type=int, default=7)
    p.add_argument("--lr-scheduler-factor", type=float, default=0.5)
    p.add_argument("--lr-scheduler-min-lr", type=float, default=1e-6)
    p.add_argument("--base-channels", type=int, default=32)
    p.add_argument("--checkpoint-dir", default="checkpoints")
    p.add_argument("--run-name", default="cae_anomaly")
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--normalizer-path", default=None)
    p.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    p.add_argument("--trainer", default="native", choices=["native", "autograd"],
                   help="native: the whole step on the C ABI (no reconstruction / loss-gradient tensors, fused AdamW); autograd: "
                        "torch criterion + loss.backward() over the C-ABI autograd bridge")
    p.add_argument("--flat-input", action="store_true",
                   help="one GPU: read the features through the flat memory-mapped file + row-gather loader of the data-parallel path "
                        "(the reference's DataLoader normalises every utterance on the host, two workers: a few thousand utterances/s; "
                        "the flat path keeps up with the 44 k utterances/s of the training step); batch order then comes from "
                        "torch.randperm(seed + epoch)")
    p.add_argument("--sync-bn", action="store_true",
                   help="data-parallel training with the native trainer: BatchNorm statistics over the global batch "
                        "(default: each rank's own statistics, as torch DistributedDataParallel)")
    return p.parse_args(argv)


def main(argv=None):
    from . import distributed as _dist
    _dist.limit_cpu_threads()      # the job's CPU share, not the machine's CPU count (distributed.cpu_budget)
    args = parse_args(argv)
    random.seed(args.seed)
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", local_rank) if str(args.device).startswith("cuda") else torch.device(args.device)
    if device.type == "cuda":
        torch.cuda.set_device(device)
    rank, world = dfa_dist.init(device=device)
    ckpt_dir = os.path.join(args.checkpoint_dir, args.run_name)
    os.makedirs(ckpt_dir, exist_ok=True)
    best_path, last_path = os.path.join(ckpt_dir, "cae_best.pt"), os.path.join(ckpt_dir, "cae_last.pt")
    flat = world > 1 or args.flat_input     # data-parallel input path: each rank reads only what it consumes
    if flat:
        # rank 0 alone converts a features.pkl into the flat memory-mapped file; every rank maps it, fits the normaliser on its
        # share of the bonafide rows (one all-reduce) and fetches exactly the rows of its batches -- no rank un-pickles or holds
        # the whole set (the classifiers' path: train.py / dataloaders.open_flat)
        from .dataloaders import IndexedFlatBatcher, open_flat
        cache = os.path.join(ckpt_dir, "flat_cache")
        tr_feats, tr_labels, _ = open_flat(args.train_features, args.train_labels, cache, rank, world, "cae_train")
        dv_feats, dv_labels, _ = open_flat(args.dev_features, args.dev_labels, cache, rank, world, "cae_dev")
        tr_rows = (tr_labels == 1).nonzero().reshape(-1)
        dv_rows = (dv_labels == 1).nonzero().reshape(-1)
        if args.normalizer_path and os.path.exists(args.normalizer_path):
            normalizer = FeatureNormalizer.load(args.normalizer_path)
        else:
            normalizer = fit_normalizer_sharded(tr_feats, tr_rows, rank, world)
            if rank == 0:
                normalizer.save(os.path.join(ckpt_dir, "normalizer.pt"))
        nmean, nstd = normalizer.mean.to(device), normalizer.std.to(device)
        n_train = int(tr_rows.numel())
    else:
        if args.normalizer_path and os.path.exists(args.normalizer_path):
            normalizer = FeatureNormalizer.load(args.normalizer_path)
        else:
            normalizer = build_normalizer(args.train_features, args.train_labels)
            normalizer.save(os.path.join(ckpt_dir, "normalizer.pt"))
        train_ds = BonafideDataset(args.train_features, args.train_labels, normalizer=normalizer, swap_tf=True)
        val_ds = BonafideDataset(args.dev_features, args.dev_labels, normalizer=normalizer, swap_tf=True)

    model = ConvAutoencoder(base_channels=args.base_channels, precision=args.precision).to(device)
    criterion = nn.MSELoss()
    sched_kw = dict(mode="min", factor=args.lr_scheduler_factor, patience=args.lr_scheduler_patience, threshold=1e-4,
                    min_lr=args.lr_scheduler_min_lr)
    if world > 1 or args.trainer == "native":
        # src/train_cae.py:58-82 on every rank's shard: one flat 2,246,532-byte gradient all-reduce per step (world > 1), fused
        # AdamW, equal step counts on every rank
        from .training.train_step import CaeNativeTrainer, FlatTrainer
        Trainer = CaeNativeTrainer if args.trainer == "native" else FlatTrainer
        kw = {"sync_bn": True} if (args.sync_bn and world > 1 and args.trainer == "native") else {}
        optimizer = Trainer(model, lr=args.lr, weight_decay=args.weight_decay, **kw)
        dfa_dist.broadcast_parameters_(optimizer.flat_p)
        scheduler = optimizer.plateau_scheduler(**sched_kw)
    else:
        optimizer = torch.optim.AdamW(model.parameters(), lr=args.lr, weight_decay=args.weight_decay)
        scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, **sched_kw)
    if not flat:
        train_loader = DataLoader(train_ds, batch_size=args.batch_size, shuffle=True, num_workers=args.num_workers)
        val_loader = DataLoader(val_ds, batch_size=args.batch_size, shuffle=False, num_workers=args.num_workers)
    best, no_improve, last_epoch = None, 0, 0
    resident = None
    for epoch in range(1, args.epochs + 1):
        if flat:
            perm = torch.randperm(n_train, generator=torch.Generator().manual_seed(args.seed + epoch))
            idx = tr_rows[train_shard_indices(perm, args.batch_size, rank, world)]     # rows of the flat source, this rank's share
            if resident is None:
                # the whole flat source in HBM when it needs at most half of the free memory (dataloaders.ResidentBatcher):
                # uploaded once, batches are device-side row gathers
                from .dataloaders import ResidentBatcher
                resident = (ResidentBatcher(tr_feats, None, args.batch_size, device=device)
                            if (device.type == "cuda" and ResidentBatcher.fits(tr_feats, device)) else False)
            batcher = resident.epoch(idx) if resident else IndexedFlatBatcher(tr_feats, None, idx, args.batch_size, device=device)
            train_loss = dfa_dist.mean_scalar(train_one_epoch(model, _NormalizedBatches(batcher, nmean, nstd, torch.bfloat16 if (args.precision == "bf16" and device.type == "cuda") else None), criterion, optimizer, device), device)
            dfa_dist.average_tensors_(dfa_dist.bn_running_stats(model))
            model._prepared = None
            val_mse = validate_reconstruction_flat(model, dv_feats, dv_rows, nmean, nstd, args.batch_size, device, rank, world)
        else:
            train_loss = train_one_epoch(model, train_loader, criterion, optimizer, device)
            val_mse = validate_reconstruction(model, val_loader, device)
        scheduler.step(val_mse)
        is_best = best is None or val_mse < best
        if rank == 0:
            print(f"epoch {epoch}: train_mse={train_loss:.6f} val_mse={val_mse:.6f}" + ("  *best*" if is_best else ""))
        if is_best:
            best, no_improve = val_mse, 0
            if rank == 0:
                save_checkpoint(model, optimizer, epoch, args, best_path, scheduler=scheduler)
        else:
            no_improve += 1
        last_epoch = epoch
        if args.early_stop and no_improve >= args.early_stop:
            break
    if rank == 0:
        save_checkpoint(model, optimizer, last_epoch, args, last_path, scheduler=scheduler)


if __name__ == "__main__":
    main()
