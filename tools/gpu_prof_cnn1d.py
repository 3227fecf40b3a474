"""Profile target: 10 CNN1D eval forwards + 4 training steps at [256,321,180] fp32 -- run under rocprofv3."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dfa_amd.model_cnn1d import CNN1D
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
x = (torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07).to(dev).transpose(1, 2)
y = (torch.rand(256, generator=g) > 0.5).float().to(dev)
torch.manual_seed(0)
m = CNN1D().to(dev).eval()
for _ in range(10): m(x)
m.train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
for _ in range(4):
    opt.zero_grad(set_to_none=True)
    loss = torch.nn.BCEWithLogitsLoss()(m(x).squeeze(-1), y)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
