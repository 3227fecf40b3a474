"""Probe: CNN2D bf16-mode training step with the augmentation FOLDED into the loads vs the stand-alone pass feeding the same
values, parameter by parameter (tests/test_train_shapes_gpu.py augment_folded cases disagreed with the oracle in conv.0.weight)."""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfa_amd import _lib
from dfa_amd.augmentation import FusedAugment
from dfa_amd.model import CNN2D
from oracle import torch_ref as R

torch.manual_seed(0)
B, T, F = 3, 40, 180
cfg = dict(spec_augment=True, time_mask_ratio=0.2, feature_mask=True, feature_mask_ratio=0.1, time_shift=True,
           time_shift_ratio=0.1, channel_drop=True, channel_drop_prob=0.3, gaussian_jitter=False, gaussian_jitter_std=0.05)
g = torch.Generator().manual_seed(1)
stored = (torch.randn(B, F, T, generator=g) * 3.2).to(torch.bfloat16)
y = (torch.rand(B, generator=g) > 0.5).float().cuda()
x16 = stored.cuda().transpose(1, 2)
ctx = _lib.Context.get(x16.device)


def run(prec, x, fold, fused_bwd=1):
    torch.manual_seed(9)
    m = CNN2D(in_features=F, dropout=0.0, precision=prec).cuda().train()
    with torch.no_grad():
        m.classifier.weight.mul_(30.0)
    ctx.set_option("conv1_bwd_fused", fused_bwd)
    random.seed(31); torch.manual_seed(31)
    aug = FusedAugment(seed=77, fold=fold, out_dtype=torch.float32, **cfg)
    xa = aug(x)
    logits = m(xa).squeeze(-1)
    torch.nn.BCEWithLogitsLoss()(logits, y).backward()
    ctx.set_option("conv1_bwd_fused", 1)
    return logits.detach(), {n: p.grad.clone() for n, p in m.named_parameters()}, xa, {k: v.detach().cpu() for k, v in m.state_dict().items()}


for prec in ("fp32", "bf16"):
    for xin, tag in ((x16, "bf16 x"), (x16.float(), "fp32 x")):
        for fb in (1, 0):
            l0, g0, xa, sd = run(prec, xin, False, fb)
            l1, g1, _, _ = run(prec, xin, True, fb)
            worst = max(((float((g0[n] - g1[n]).abs().max()) / max(float(g0[n].abs().max()), 1e-9)), n) for n in g0)
            print(f"{prec} mode, {tag}, conv1_bwd_fused={fb}: logits diff {float((l0 - l1).abs().max()):.2e}; worst grad diff fold vs stand-alone: {worst[0]:.2e} ({worst[1]})")
# stand-alone path vs the emulated oracle (is the stand-alone one right?)
l0, g0, xa, sd = run("bf16", x16, False)
torch.manual_seed(9)
m = CNN2D(in_features=F, dropout=0.0)
with torch.no_grad():
    m.classifier.weight.mul_(30.0)
sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
_, _, emu = R.cnn2d_train_step_emulated(sd0, xa.float().cpu(), y.cpu(), 0.0, "bf16")
for n in g0:
    if n in ("conv.0.bias", "conv.5.bias", "conv.10.bias"):
        continue
    sc = max(float(emu[n].abs().max()), 1e-9)
    print(f"   stand-alone bf16 vs emulated oracle {n}: {float((g0[n].cpu() - emu[n]).abs().max()) / sc:.2e}")
l1, g1, _, _ = run("bf16", x16, True)
for n in ("conv.0.weight", "conv.1.weight", "conv.1.bias", "conv.5.weight"):
    sc = max(float(emu[n].abs().max()), 1e-9)
    print(f"   folded bf16 vs emulated oracle {n}: {float((g1[n].cpu() - emu[n]).abs().max()) / sc:.2e}")
