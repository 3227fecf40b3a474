"""Profile target: 10 bf16 CAE score passes (fused z-score + reconstruction + MSE) at [256,321,180] -- run under rocprofv3."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dfa_amd.model_cae import ConvAutoencoder
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
x = (torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07).to(dev, dtype=torch.bfloat16).transpose(1, 2)
torch.manual_seed(0)
cae = ConvAutoencoder(precision="bf16").to(dev).eval()
mean, std = torch.zeros(180, device=dev), torch.ones(180, device=dev)
for _ in range(10): cae.score(x, mean, std)
torch.cuda.synchronize()
