import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dfa_amd.model import CNN2D
from dfa_amd.training.train_step import NativeTrainer
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator().manual_seed(1)
stored = torch.randn(B, 180, 321, generator=g) * 3.2 - 0.07
y = (torch.rand(B, generator=g) > 0.5).float().to(dev)
for prec in ("bf16", "fp32"):
    torch.manual_seed(0)
    model = CNN2D(dropout=0.2, precision=prec).to(dev)
    x = (stored.to(dev, dtype=torch.bfloat16) if prec == "bf16" else stored.to(dev)).transpose(1, 2)
    tr = NativeTrainer(model, label_smoothing=0.05)
    for _ in range(3): tr.step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): loss = tr.step(x, y)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{prec}: train step {dt*1e3:.2f} ms  -> {B/dt:.0f} utt/s  loss {loss.item():.4f}", flush=True)
