"""Probe: CNN1D training gradients with the bf16x3 layer kernel (cnn1d_train_x3 = 1, 2) against the fp32 VALU convolutions (0),
parameter by parameter, same weights and batch, dropout off."""
import sys, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dfa_amd import _lib
from dfa_amd.model_cnn1d import CNN1D
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
B, T, F = 3, 321, 180
x = (torch.randn(B, F, T, generator=g) * 3.2).to(dev).transpose(1, 2)
y = (torch.rand(B, generator=g) > 0.5).float().to(dev)
ctx = _lib.Context.get(dev)
res = {}
for arm in (0, 1, 2, 3):
    ctx.set_option("cnn1d_train_x3", arm)
    torch.manual_seed(0)
    m = CNN1D(in_features=F, dropout=0.0).to(dev).train()
    logits = m(x).squeeze(-1)
    torch.nn.BCEWithLogitsLoss()(logits, y).backward()
    res[arm] = (logits.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters()})
ctx.set_option("cnn1d_train_x3", 1)
for arm in (1, 2, 3):
    print(f"arm {arm}: logits max diff {float((res[arm][0] - res[0][0]).abs().max()):.3e}")
    for n in res[0][1]:
        a, b = res[0][1][n], res[arm][1][n]
        sc = float(a.abs().max()) + 1e-30
        print(f"   {n:22s} max {float((a - b).abs().max()) / sc:.3e}  relL2 {float((a - b).norm() / (a.norm() + 1e-30)):.3e}")
