"""Diagnostic (GPU box): the asm-pipelined fp32 ACCIN data-gradient kernel (train_conv_variant 7) against its
compiler-scheduled twin; prints WHERE the outputs differ (rows / columns / channels of da2) to localise the cause."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from dfa_amd import _lib
from dfa_amd.model import CNN2D

def al(v): return (v + 255) // 256 * 256
def run(variant, B, T, F=180):
    ctx = _lib.Context.get(torch.device("cuda"))
    ctx.set_option("train_conv_variant", variant)
    torch.manual_seed(3)
    m = CNN2D(in_features=F, dropout=0.0, precision="fp32").to("cuda").train()
    with torch.no_grad(): m.classifier.weight.mul_(30.0)
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(B, F, T, generator=g) * 3.2).to("cuda").transpose(1, 2)
    y = (torch.rand(B, generator=g) > 0.5).float().to("cuda")
    loss = torch.nn.BCEWithLogitsLoss()(m(x).squeeze(-1), y); loss.backward()
    torch.cuda.synchronize()
    H1, H2 = T // 2, T // 4
    off = 0
    sizes = [B*H1*F*32*4, B*H1*F*64*4, B*H2*F*64*4, B*H2*F*128*4, B*128*F*4, B*128*F*4, B*128*F*8, B*H2*F*128*4]
    for sz in sizes: off = al(off + sz)
    da2 = m._train_ws[off: off + B*H2*F*64*4].view(torch.float32).view(B, H2, F, 64).clone()
    grads = {n: p.grad.clone() for n, p in m.named_parameters()}
    return da2, grads
for (B, T) in ((2, 64), (4, 321)):
    d0, g0 = run(2, B, T)
    d7, g7 = run(7, B, T)
    diff = (d0 - d7).abs()
    print(f"B={B} T={T}: da2 max|diff| {float(diff.max()):.3e} of scale {float(d0.abs().max()):.3e}; differing elements {int((diff > 0).sum())} of {diff.numel()}")
    if float(diff.max()) > 0:
        bad = (diff > 1e-6 * float(d0.abs().max()))
        print("  rows (t) with differences:", torch.nonzero(bad.any(dim=3).any(dim=2).any(dim=0)).flatten().tolist()[:80])
        print("  cols (f) with differences:", torch.nonzero(bad.any(dim=3).any(dim=1).any(dim=0)).flatten().tolist()[:200])
        print("  channels with differences:", torch.nonzero(bad.any(dim=2).any(dim=1).any(dim=0)).flatten().tolist())
        print("  utterances:", torch.nonzero(bad.any(dim=3).any(dim=2).any(dim=1)).flatten().tolist())
        idx = torch.nonzero(bad)[:10]
        for i in idx:
            b, t, f, c = [int(v) for v in i]
            print("   ", (b, t, f, c), float(d0[b, t, f, c]), float(d7[b, t, f, c]))
_lib.Context.get(torch.device("cuda")).set_option("train_conv_variant", 2)

# ---- the CAE encoder block 4 (fp32, Cin split): <float, 64, 4, 1, 1, 1, EPI_POOL_2X2, 1, ACCIN> with 288 weight registers
from dfa_amd.model_cae import ConvAutoencoder
ctx = _lib.Context.get(torch.device("cuda"))
torch.manual_seed(1)
cae = ConvAutoencoder(precision="fp32").to("cuda").eval()
for (B, T) in ((2, 64), (3, 321), (64, 321)):
    g = torch.Generator().manual_seed(B)
    x = torch.randn(B, T, 180, generator=g).to("cuda")
    outs = {}
    for v in (2, 7, 8):
        ctx.set_option("train_conv_variant", v)
        recon, latent = cae(x)
        outs[v] = latent.clone()
    ctx.set_option("train_conv_variant", 2)
    for v in (7, 8):
        diff = (outs[2] - outs[v]).abs()
        print(f"CAE enc4 B={B} T={T} variant {v}: latent max|diff| {float(diff.max()):.3e} of scale {float(outs[2].abs().max()):.3e}; "
              f"differing {int((diff > 0).sum())} of {diff.numel()}")
        if float(diff.max()) > 0:
            bad = diff > 1e-6 * float(outs[2].abs().max())
            print("  channels:", torch.nonzero(bad.any(dim=3).any(dim=2).any(dim=0)).flatten().tolist()[:64])
            print("  rows:", torch.nonzero(bad.any(dim=3).any(dim=1).any(dim=0)).flatten().tolist())
            print("  cols:", torch.nonzero(bad.any(dim=2).any(dim=1).any(dim=0)).flatten().tolist())
            print("  utterances:", torch.nonzero(bad.any(dim=3).any(dim=2).any(dim=1)).flatten().tolist()[:64])
            for i in torch.nonzero(bad)[:8]:
                b, c, t, f = [int(q) for q in i]
                print("   ", (b, c, t, f), float(outs[2][b, c, t, f]), float(outs[v][b, c, t, f]))
