"""CNN1D training-step throughput (autograd bridge + torch AdamW) at B=256."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dfa_amd.model_cnn1d import CNN1D
dev = torch.device("cuda", 0)
B = 256
g = torch.Generator().manual_seed(1)
x = (torch.randn(B, 180, 321, generator=g) * 3.2 - 0.07).to(dev).transpose(1, 2)
y = (torch.rand(B, generator=g) > 0.5).float().to(dev)
torch.manual_seed(0)
m = CNN1D(dropout=0.2).to(dev).train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
crit = torch.nn.BCEWithLogitsLoss()
def step():
    loss = crit(m(x).squeeze(-1), y)
    opt.zero_grad(); loss.backward(); opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n): loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"cnn1d fp32: train step {dt*1e3:.2f} ms -> {B/dt:.0f} utt/s loss {loss.item():.4f}", flush=True)
