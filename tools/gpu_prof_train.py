"""Profile target: 6 bf16 CNN2D training steps at [256,321,180] (NativeTrainer) -- run under rocprofv3."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dfa_amd.model import CNN2D
from dfa_amd.training.train_step import NativeTrainer
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
stored = torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07
y = (torch.rand(256, generator=g) > 0.5).float().to(dev)
torch.manual_seed(0)
model = CNN2D(dropout=float(__import__("os").environ.get("DFA_PROF_DROPOUT", "0.2")), precision="bf16").to(dev)
x = stored.to(dev, dtype=torch.bfloat16).transpose(1, 2)
tr = NativeTrainer(model, label_smoothing=0.05)
for _ in range(6): tr.step(x, y)
torch.cuda.synchronize()
