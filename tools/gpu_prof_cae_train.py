"""Profile target: 6 bf16 ConvAutoencoder training steps at B=256, T=321 (tools/gpu_profile_one.sh cae_train tools/gpu_prof_cae_train.py 6)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfa_amd.model_cae import ConvAutoencoder
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
x32 = torch.randn(256, 321, 180, generator=g).to(dev)
x = x32.to(torch.bfloat16)
torch.manual_seed(0)
m = ConvAutoencoder(precision="bf16").to(dev).train()
opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
for _ in range(6):
    recon, _ = m(x)
    loss = torch.nn.functional.mse_loss(recon, x32)
    opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
