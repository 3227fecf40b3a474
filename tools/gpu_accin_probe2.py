"""Diagnostic (GPU box): localise the wrong sums of the pipelined (3 reads in flight) fp32 ACCIN kernel (CAE encoder block 4,
second Cin half) by switching on ONE (tap, 8-channel k-group) of the weights at a time and comparing variant 7 with the
compiler-scheduled kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from dfa_amd import _lib
from dfa_amd.model_cae import ConvAutoencoder
ctx = _lib.Context.get(torch.device("cuda"))
torch.manual_seed(1)
cae = ConvAutoencoder(precision="fp32").to("cuda").eval()
w_full = cae.encoder[12].weight.detach().clone()          # [256, 128, 3, 3]
g = torch.Generator().manual_seed(2)
x = torch.randn(2, 112, 180, generator=g).to("cuda")      # 7 latent rows: ring phases 0, 1, 2, 0, 1, 2, 0
bad = {}
for tap in range(9):
    for kg in range(8):
        w = torch.zeros_like(w_full)
        ci0 = 64 + 8 * kg
        w[:, ci0:ci0 + 8, tap // 3, tap % 3] = w_full[:, ci0:ci0 + 8, tap // 3, tap % 3]
        with torch.no_grad():
            cae.encoder[12].weight.copy_(w)
        outs = {}
        for v in (2, 7):
            ctx.set_option("train_conv_variant", v)
            outs[v] = cae(x)[1].clone()
        d = (outs[2] - outs[7]).abs()
        if float(d.max()) > 0:
            rows = torch.nonzero(d.amax(dim=(0, 1, 3)) > 0).flatten().tolist()
            chans = torch.nonzero(d.amax(dim=(0, 2, 3)) > 0).flatten().tolist()
            bad[(tap, kg)] = (float(d.max()), float(outs[2].abs().max()), rows, len(chans), chans[:12])
ctx.set_option("train_conv_variant", 2)
print("differing (tap, k-group) combinations:", len(bad), "of 72")
for k, v in sorted(bad.items()):
    print(k, "max|diff| %.3e of %.3e" % v[:2], "latent rows", v[2], "channels", v[3], v[4])
