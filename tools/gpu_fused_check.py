"""Compare the fused block-1+2 kernel against the two-kernel path and the fp32 oracle on bf16 features."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_golden
from dfa_amd import _lib
from dfa_amd.model import CNN2D
from oracle import dfa_oracle as O
sd, g = load_golden("cnn2d_eval")
dev = torch.device("cuda")
ctx = _lib.Context.get(dev)
def mk(F):
    m = CNN2D(in_features=F, precision="bf16")
    if F == 180: m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.to(dev).eval()
gen = torch.Generator().manual_seed(5)
cases = [("t321", torch.from_numpy(g["t321.x_stored"]).transpose(1, 2)), ("t7", torch.from_numpy(g["t7.x_stored"]).transpose(1, 2)),
         ("tf", torch.randn(3, 50, 180, generator=gen)),            # [B,T,F] contiguous (f fastest)
         ("F65", torch.randn(2, 33, 65, generator=gen)), ("F40", torch.randn(2, 18, 40, generator=gen)),
         ("big", (torch.randn(64, 180, 321, generator=gen) * 3).transpose(1, 2))]
for name, x in cases:
    F = x.shape[2]
    m = mk(F)
    xb = x.to(dev).to(torch.bfloat16) if name == "tf" else x.transpose(1, 2).contiguous().to(dev).to(torch.bfloat16).transpose(1, 2)
    ctx.set_option("fuse_conv1", 0); l0, e0 = m(xb, return_embedding=True)
    ctx.set_option("fuse_conv1", 1); l1, e1 = m(xb, return_embedding=True)
    msg = f"{name:5s} shape {tuple(xb.shape)} strides {xb.stride()} | logits max|d| {(l0-l1).abs().max().item():.3e} (scale {l0.abs().max().item():.3e}) emb max|d| {(e0-e1).abs().max().item():.3e} (scale {e0.abs().max().item():.3e})"
    if F == 180 and x.shape[0] <= 8:
        ref = O.cnn2d_forward({k: np.asarray(v) for k, v in sd.items()}, xb.float().cpu().numpy())
        ref = ref[0] if isinstance(ref, tuple) else ref
        msg += f" | vs fp32 oracle: unfused {np.abs(l0.cpu().numpy().reshape(-1) - ref.reshape(-1)).max():.3e} fused {np.abs(l1.cpu().numpy().reshape(-1) - ref.reshape(-1)).max():.3e}"
    print(msg, flush=True)
ctx.set_option("fuse_conv1", 1)
