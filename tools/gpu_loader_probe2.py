"""Probe: components of the row-gather loader -- index_select of 256 random rows of a memory-mapped [N,180,321] file into a
preallocated pinned buffer at several thread counts, fancy indexing + pin_memory() (what IndexedFlatBatcher did), and the H2D."""
import os, sys, time, tempfile, warnings
import numpy as np, torch
N, B = 4096, 256
d = tempfile.mkdtemp()
dev = torch.device("cuda", 0)
path = os.path.join(d, "f.npy")
arr = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=(N, 180, 321))
arr[:] = 1.0
arr.flush(); del arr
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    t = torch.from_numpy(np.load(path, mmap_mode="r"))
_ = float(t.sum())
perm = torch.randperm(N)
pinned = torch.empty(B, 180, 321, dtype=torch.float32).pin_memory()
def tm(fn, n=8):
    fn(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    return (time.perf_counter() - t0) / n * 1e3
for nt in (1, 4, 8, 16, 32):
    torch.set_num_threads(nt)
    ms = tm(lambda i=0: torch.index_select(t, 0, perm[(i % 15) * B:(i % 15 + 1) * B], out=pinned))
    print(f"index_select -> pinned out, {nt} threads: {ms:.2f} ms per batch ({B * 231120 / ms / 1e6:.1f} GB/s)", flush=True)
torch.set_num_threads(8)
ms = tm(lambda i=0: t[perm[(i % 15) * B:(i % 15 + 1) * B]])
print(f"fancy index (new tensor), 8 threads: {ms:.2f} ms")
ms = tm(lambda i=0: t[perm[(i % 15) * B:(i % 15 + 1) * B]].pin_memory())
print(f"fancy index + pin_memory(), 8 threads: {ms:.2f} ms")
import ctypes
def rowcopy(i=0):
    idx = perm[(i % 15) * B:(i % 15 + 1) * B].tolist()
    for k, r in enumerate(idx): pinned[k].copy_(t[r])
ms = tm(rowcopy, 3)
print(f"row-by-row copy_ into pinned: {ms:.2f} ms")
torch.cuda.synchronize()
ms = tm(lambda i=0: (pinned.to(dev, non_blocking=True), torch.cuda.synchronize()))
print(f"H2D of the pinned batch: {ms:.2f} ms")
