import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
stored = torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07
x = stored.to(device=dev, dtype=torch.bfloat16).transpose(1, 2)
model = bench.build_model(torch, dev, "bf16")
for _ in range(4): model(x)
torch.cuda.synchronize()
