"""Probe: H2D bandwidth of a [N,180,321] bf16 feature file -- pageable memmap (what FlatBatcher feeds today), the same mapping
registered with the HIP runtime (hipHostRegister: direct DMA from the page cache, no staging copy), and a pinned staging buffer."""
import os, sys, time, tempfile
import numpy as np, torch
N = 2048
d = tempfile.mkdtemp()
path = os.path.join(d, "f.npy")
arr = np.lib.format.open_memmap(path, mode="w+", dtype=np.uint16, shape=(N, 180, 321))
arr[:] = np.random.randint(0, 65535, size=(N, 180, 321), dtype=np.uint16)
arr.flush(); del arr
dev = torch.device("cuda", 0)
def bench(t, label, bs=256, reps=3):
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        outs = [t[i:i + bs].to(dev, non_blocking=True) for i in range(0, N, bs)]
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    gb = t.numel() * 2 / 1e9
    print(f"{label}: {gb / best:.1f} GB/s  ({N / best / 1e3:.0f} k utt/s)", flush=True)
for mode in ("r", "c"):
    m = np.load(path, mmap_mode=mode)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t = torch.from_numpy(m).view(torch.bfloat16)
    _ = float(t.view(torch.int16).float().sum())            # touch every page (page cache resident)
    bench(t, f"pageable memmap mode={mode}")
    rt = torch.cuda.cudart()
    for flags in (0, 8):                                      # 8 = hipHostRegisterReadOnly (if the runtime knows it)
        try:
            rc = rt.cudaHostRegister(t.data_ptr(), t.numel() * 2, flags)
            print(f"  hipHostRegister(mode={mode}, flags={flags}) -> {rc}")
            if int(rc) == 0:
                print("  is_pinned:", t.is_pinned())
                bench(t, f"registered memmap mode={mode} flags={flags}")
                rt.cudaHostUnregister(t.data_ptr())
                break
        except Exception as e:  # noqa: BLE001
            print(f"  hipHostRegister(mode={mode}, flags={flags}) raised {type(e).__name__}: {e}")
pin = torch.empty(N, 180, 321, dtype=torch.bfloat16).pin_memory()
pin.copy_(t)
bench(pin, "pinned tensor (upper bound)")
