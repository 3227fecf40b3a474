"""Batch-size sweep of the CNN2D eval forward (GPU box): utterances/s per mode with the automatic time-axis split and without."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from dfa_amd import _lib
dev = torch.device("cuda", 0)
ctx = _lib.Context.get(dev)
g = torch.Generator().manual_seed(1)
stored = torch.randn(1024, 180, 321, generator=g) * 3.2 - 0.07
for prec in ("bf16", "bf16x3"):
    model = bench.build_model(torch, dev, prec)
    xs = stored.to(dev, dtype=torch.bfloat16 if prec == "bf16" else torch.float32)
    for B in (1, 8, 16, 32, 64, 128, 256):
        x = xs[:B].transpose(1, 2)
        row = []
        for split in (0, -1):
            ctx.set_option("time_split", split)
            for _ in range(20): model(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 200
            for _ in range(n): model(x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            row.append((dt * 1e3, B / dt))
        print(f"{prec} B={B:4d}: no split {row[0][0]:.3f} ms {row[0][1]:9.0f} utt/s | auto split {row[1][0]:.3f} ms {row[1][1]:9.0f} utt/s", flush=True)
ctx.set_option("time_split", -1)
