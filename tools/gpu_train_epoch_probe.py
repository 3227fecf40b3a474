"""Probe: one training epoch as `train.py --native` runs it (flat memory-mapped file -> IndexedFlatBatcher -> NativeTrainer.step),
4096 synthetic utterances, batch 256: utterances/s of the whole loop against the step alone."""
import os, sys, time, tempfile, warnings
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfa_amd.dataloaders import IndexedFlatBatcher, ResidentBatcher
from dfa_amd.model import CNN2D
from dfa_amd.model_cnn1d import CNN1D
from dfa_amd.training.train_step import NativeTrainer
N, B = 4096, 256
dev = torch.device("cuda", 0)
path = os.path.join(tempfile.mkdtemp(), "f.npy")
arr = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(N, 180, 321))
arr[:] = (np.random.randn(N, 180, 321) * 3).astype(np.float32)
arr.flush(); del arr
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    feats = torch.from_numpy(np.load(path, mmap_mode="r"))
_ = float(feats.sum())
labels = (torch.rand(N) > 0.5).float()
for name, model, dt, cast in (("cnn2d bf16, cast on the copy stream", CNN2D(precision="bf16", dropout=0.3), torch.bfloat16, False),
                              ("cnn2d bf16, fp32 feed", CNN2D(precision="bf16", dropout=0.3), None, False),
                              ("cnn2d bf16, fp32 feed + cast on the main stream", CNN2D(precision="bf16", dropout=0.3), None, True),
                              ("cnn1d", CNN1D(dropout=0.2), None, False)):
    torch.manual_seed(0)
    tr = NativeTrainer(model.to(dev), label_smoothing=0.05)
    for ep in range(2):
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(ep))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for f, l in IndexedFlatBatcher(feats, labels, perm, B, device=dev, dtype=dt):
            if cast: f = f.to(torch.bfloat16)
            loss = tr.step(f.transpose(1, 2), l)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    # the same epoch with the set resident in HBM (uploaded once, device-side row gathers)
    t0 = time.perf_counter()
    res = ResidentBatcher(feats, labels, B, device=dev, dtype=(torch.bfloat16 if (dt is not None or cast) else None))
    torch.cuda.synchronize(); up = time.perf_counter() - t0
    for ep in range(2):
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(ep))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for f, l in res.epoch(perm):
            loss = tr.step(f.transpose(1, 2), l)
        torch.cuda.synchronize(); elr = time.perf_counter() - t0
    print(f"{name}: resident set ({res.bytes_resident / 1e9:.2f} GB uploaded in {up:.2f} s): epoch loop {N / elr / 1e3:.1f} k utt/s ({elr / (N / B) * 1e3:.2f} ms per batch)", flush=True)
    del res
    x = f.transpose(1, 2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(16): tr.step(x, l)
    torch.cuda.synchronize(); st = (time.perf_counter() - t0) / 16
    print(f"{name}: epoch loop {N / el / 1e3:.1f} k utt/s ({el / (N / B) * 1e3:.2f} ms per batch); step alone {B / st / 1e3:.1f} k utt/s ({st * 1e3:.2f} ms)", flush=True)
