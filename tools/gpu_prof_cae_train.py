"""Profile target: 6 bf16 ConvAutoencoder training steps at B=256, T=321 on the all-C-ABI trainer (CaeNativeTrainer: no
reconstruction / loss-gradient tensors, no torch elementwise kernels) -- run under rocprofv3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfa_amd.model_cae import ConvAutoencoder
from dfa_amd.training.train_step import CaeNativeTrainer
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
x = torch.randn(256, 321, 180, generator=g).to(dev).to(torch.bfloat16)
torch.manual_seed(0)
tr = CaeNativeTrainer(ConvAutoencoder(precision="bf16").to(dev).train(), lr=1e-4, weight_decay=1e-4)
for _ in range(6): tr.step(x)
torch.cuda.synchronize()
