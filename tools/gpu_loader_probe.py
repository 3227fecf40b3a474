"""Probe: how fast does the data-parallel training loader (IndexedFlatBatcher: random rows of the flat memory-mapped file ->
pinned staging -> H2D) deliver batches, against the 4.7 ms / 256 utterances the CNN2D training step consumes?"""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfa_amd.dataloaders import IndexedFlatBatcher
N, B = 4096, 256
d = tempfile.mkdtemp()
dev = torch.device("cuda", 0)
for dt, npdt in (("fp32", np.float32), ("bf16", np.uint16)):
    path = os.path.join(d, f"f_{dt}.npy")
    arr = np.lib.format.open_memmap(path, mode="w+", dtype=npdt, shape=(N, 180, 321))
    arr[:] = (np.random.randn(N, 180, 321) * 3).astype(np.float32) if dt == "fp32" else np.random.randint(0, 30000, size=(N, 180, 321), dtype=np.uint16)
    arr.flush(); del arr
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t = torch.from_numpy(np.load(path, mmap_mode="r"))
    if dt == "bf16": t = t.view(torch.bfloat16)
    _ = float(t[:, 0, 0].float().sum())
    labels = torch.zeros(N)
    perm = torch.randperm(N, generator=torch.Generator().manual_seed(0))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); nb = 0
        for f, l in IndexedFlatBatcher(t, labels, perm, B, device=dev):
            nb += 1
        torch.cuda.synchronize(); el = time.perf_counter() - t0
    print(f"{dt} rows: {el / nb * 1e3:.2f} ms per batch of {B} -> {N / el / 1e3:.0f} k utt/s ({t[0].numel() * t.element_size() * N / el / 1e9:.1f} GB/s)  [torch threads {torch.get_num_threads()}]", flush=True)
