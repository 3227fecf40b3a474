"""Profile target: 6 CNN2D eval forwards per precision mode at [256,321,180] -- run under rocprofv3 (kernel trace or --pmc)."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
stored = torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07
x16 = stored.to(device=dev, dtype=torch.bfloat16).transpose(1, 2)
x32 = stored.to(device=dev).transpose(1, 2)
modes = sys.argv[1:] or ["bf16", "bf16x3"]
for prec in modes:
    model = bench.build_model(torch, dev, prec)
    x = x16 if prec == "bf16" else x32
    for _ in range(6): model(x)
    torch.cuda.synchronize()
