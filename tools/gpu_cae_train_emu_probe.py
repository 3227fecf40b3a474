"""Calibration probe (GPU box): bf16-mode auto-encoder training gradients vs the rounding-faithful training oracle and vs the
fp32 reference goldens, at the golden batch and at a larger one."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from oracle import torch_ref as R
from dfa_amd.model_cae import ConvAutoencoder
_, g = load_golden("cae_train")
sd = {k[len("init.sd."):]: v for k, v in g.items() if k.startswith("init.sd.")}
init = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
NOISE = {f"encoder.{i}.bias" for i in (0, 4, 8, 12)} | {f"decoder.{i}.bias" for i in (0, 3, 6)}
def gpu_grads(x):
    m = ConvAutoencoder(precision="bf16"); m.load_state_dict(init); m = m.to("cuda").train()
    xb = x.to("cuda").to(torch.bfloat16)
    recon, _ = m(xb)
    torch.nn.MSELoss()(recon.float(), xb.float()).backward()
    return {n: p.grad.float().cpu() for n, p in m.named_parameters()}
gen = torch.Generator().manual_seed(4)
cases = [("golden 2x32", torch.from_numpy(g["ls0.x"])), ("24x96", torch.randn(24, 96, 180, generator=gen)), ("8x321", torch.randn(8, 321, 180, generator=gen) * 2.0)]
for name, x in cases:
    got = gpu_grads(x)
    _, emu = R.cae_train_step_emulated(sd, x, "bf16")
    _, ref = R.cae_train_step_emulated(sd, x, None)
    print(name)
    for n in got:
        if n in NOISE: continue
        s = max(float(ref[n].abs().max()), 1e-9)
        print(f"  {n:20s} vs emulated {float((got[n]-emu[n]).abs().max())/s:.4f}   vs fp32 {float((got[n]-ref[n]).abs().max())/s:.4f}   emulated vs fp32 {float((emu[n]-ref[n]).abs().max())/s:.4f}", flush=True)
