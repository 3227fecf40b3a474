import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dfa_amd.model import CNN2D
dev = torch.device("cuda", 0)
B = 256
g = torch.Generator().manual_seed(1)
x = (torch.randn(B, 180, 321, generator=g) * 3.2 - 0.07).to(dev, dtype=torch.bfloat16).transpose(1, 2)
y = (torch.rand(B, generator=g) > 0.5).float().to(dev)
torch.manual_seed(0)
m = CNN2D(dropout=0.2, precision="bf16").to(dev).train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
crit = torch.nn.BCEWithLogitsLoss()
def step():
    loss = crit(m(x).squeeze(-1), y * 0.95 + 0.025)
    opt.zero_grad(); loss.backward(); opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(f"autograd-bridge bf16 train step {dt*1e3:.2f} ms -> {B/dt:.0f} utt/s")
