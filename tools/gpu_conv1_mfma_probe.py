"""conv1 matrix-core backward vs the vector kernel on the SAME forward state, both against a float64 evaluation of the same
formulas from the workspace's own da1 (diagnostic): forward once, backward twice."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import torch.nn.functional as Fn
from dfa_amd import _lib
from dfa_amd.model import CNN2D
from dfa_amd.training.train_step import cnn2d_forward_train_raw, cnn2d_backward_raw
ctx = _lib.Context.get(torch.device("cuda"))
gen = torch.Generator().manual_seed(29)
cases = [((torch.randn(3, 65, 21, generator=gen) * 3.2 - 0.07).transpose(1, 2), torch.tensor([0.0, 1.0, 1.0])),
         ((torch.randn(16, 180, 321, generator=gen) * 3.2 - 0.07).transpose(1, 2), (torch.rand(16, generator=gen) > 0.5).float()),
         ((torch.randn(96, 180, 321, generator=gen) * 3.2 - 0.07).transpose(1, 2), (torch.rand(96, generator=gen) > 0.5).float())]
al = lambda v: (v + 255) // 256 * 256
for x, y in cases:
    B, T, F = x.shape
    H1, H2 = T // 2, T // 4
    ctx.set_option("conv1_mfma", 1)
    torch.manual_seed(4)
    model = CNN2D(in_features=F, dropout=0.0, precision="bf16").to("cuda").train()
    xb = x.to("cuda").to(torch.bfloat16)
    logits, c, ws = cnn2d_forward_train_raw(model, xb)
    dl = (torch.sigmoid(logits) - y.to("cuda").view(-1, 1)) / B
    names = [n for n, _ in model.named_parameters()]
    ga = [torch.zeros_like(p) for p in model.parameters()]
    gb = [torch.zeros_like(p) for p in model.parameters()]
    cnn2d_backward_raw(model, xb, dl.contiguous(), ga, c, ws)
    ctx.set_option("conv1_mfma", 0)
    cnn2d_backward_raw(model, xb, dl.contiguous(), gb, c, ws)
    ctx.set_option("conv1_mfma", 1)
    sizes = [B*H1*F*32*2, B*H1*F*64*2, B*H2*F*64*2, B*H2*F*128*2, B*128*F*4, B*128*F*4, B*128*F*2*4, B*H2*F*128*2, B*H2*F*64*2, B*H1*F*64*2]
    off = 0
    for s in sizes: off = al(off + s)
    da1 = ws[off: off + B*H1*F*32*2].view(torch.bfloat16).view(B, H1, F, 32).double()
    # float64 evaluation
    w = model.conv[0].weight.detach().double(); bb = model.conv[0].bias.detach().double()
    gm = model.conv[1].weight.detach().double(); bt = model.conv[1].bias.detach().double()
    xd = xb.double().unsqueeze(1)                                  # [B,1,T,F]
    z = Fn.conv2d(xd, w, bb, padding=1)                            # [B,32,T,F]
    mu = z.mean(dim=(0, 2, 3), keepdim=True); var = z.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    xh = (z - mu) / torch.sqrt(var + 1e-5)
    yv = gm.view(1, -1, 1, 1) * xh + bt.view(1, -1, 1, 1)
    g = torch.zeros_like(z)
    up = da1.permute(0, 3, 1, 2)                                   # [B,32,H1,F]
    g[:, :, 0:2 * H1:2] = 0.5 * up; g[:, :, 1:2 * H1:2] = 0.5 * up
    dy = g * (yv > 0)
    N = B * T * F
    S1 = dy.sum(dim=(0, 2, 3)); S2 = (dy * xh).sum(dim=(0, 2, 3))
    dz = (gm / torch.sqrt(var.view(-1) + 1e-5)).view(1, -1, 1, 1) * (dy - S1.view(1, -1, 1, 1) / N - xh * S2.view(1, -1, 1, 1) / N)
    dW = torch.nn.grad.conv2d_weight(xd, w.shape, dz, padding=1)
    truth = {"conv.0.weight": dW, "conv.1.weight": S2, "conv.1.bias": S1}
    print(tuple(x.shape))
    for n, a, b in zip(names, ga, gb):
        if n in truth:
            t = truth[n].float()
            scale = max(float(t.abs().max()), 1e-12)
            print("   %-16s scale %.3e  matrix-core vs f64 %.3e   vector vs f64 %.3e   between %.3e" %
                  (n, scale, float((a - t).abs().max()) / scale, float((b - t).abs().max()) / scale, float((a - b).abs().max()) / scale), flush=True)
    del z, xh, yv, g, dy, dz
