"""Profile target: 6 CNN1D training steps at [256,321,180] fp32 on the all-C-ABI trainer -- run under rocprofv3."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dfa_amd.model_cnn1d import CNN1D
from dfa_amd.training.train_step import NativeTrainer
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
x = (torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07).to(dev).transpose(1, 2)
y = (torch.rand(256, generator=g) > 0.5).float().to(dev)
torch.manual_seed(0)
tr = NativeTrainer(CNN1D(dropout=0.2).to(dev), label_smoothing=0.05)
for _ in range(6): tr.step(x, y)
torch.cuda.synchronize()
