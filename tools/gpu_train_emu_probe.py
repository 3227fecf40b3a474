"""Calibration probe (GPU box): bf16-mode CNN2D training gradients vs the rounding-faithful training oracle and vs the fp32
reference goldens, at the golden batch and at a larger one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from oracle import torch_ref as R
from dfa_amd.model import CNN2D
_, g = load_golden("cnn2d_train")
sd = {k[len("init.sd."):]: v for k, v in g.items() if k.startswith("init.sd.")}
NOISE = ("conv.0.bias", "conv.5.bias", "conv.10.bias")
def gpu_grads(x, y, prec="bf16"):
    m = CNN2D(in_features=180, dropout=0.0, precision=prec)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.to("cuda").train()
    loss = torch.nn.BCEWithLogitsLoss()(m(x.to("cuda")).squeeze(-1), y.to("cuda"))
    loss.backward()
    return {n: p.grad.float().cpu() for n, p in m.named_parameters()}
gen = torch.Generator().manual_seed(3)
cases = [("golden 4x16", torch.from_numpy(g["ls0.x"]).transpose(1, 2), torch.from_numpy(g["ls0.y"])),
         ("16x64", (torch.randn(16, 180, 64, generator=gen) * 3.2 - 0.07).transpose(1, 2), (torch.rand(16, generator=gen) > 0.5).float())]
for name, x, y in cases:
    got = gpu_grads(x.to(torch.bfloat16), y)
    _, _, emu = R.cnn2d_train_step_emulated(sd, x, y, 0.0, "bf16")
    _, _, ref = R.cnn2d_train_step_emulated(sd, x, y, 0.0, None)
    print(name)
    for n in got:
        if n in NOISE: continue
        s = max(float(ref[n].abs().max()), 1e-9)
        print(f"  {n:20s} vs emulated {float((got[n]-emu[n]).abs().max())/s:.4f}   vs fp32 {float((got[n]-ref[n]).abs().max())/s:.4f}   emulated vs fp32 {float((emu[n]-ref[n]).abs().max())/s:.4f}")
