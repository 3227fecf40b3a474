"""Throughput of the secondary paths (CNN1D, CAE score / full forward) at B=256."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dfa_amd.model_cnn1d import CNN1D
from dfa_amd.model_cae import ConvAutoencoder
dev = torch.device("cuda", 0)
B = 256
g = torch.Generator().manual_seed(1)
stored = (torch.randn(B, 180, 321, generator=g) * 3.2 - 0.07).to(dev)
x = stored.transpose(1, 2)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
torch.manual_seed(0)
m1 = CNN1D().to(dev).eval()
dt = timeit(lambda: m1(x)); print(f"cnn1d fp32  fwd: {dt*1e3:.3f} ms -> {B/dt:.0f} utt/s", flush=True)
mean, std = torch.zeros(180, device=dev), torch.ones(180, device=dev)
for prec in ("bf16", "fp32"):
    cae = ConvAutoencoder(precision=prec).to(dev).eval()
    xx = x.to(torch.bfloat16) if prec == "bf16" else x
    dt = timeit(lambda: cae.score(xx, mean, std)); print(f"cae {prec} score (fused z-score+MSE): {dt*1e3:.3f} ms -> {B/dt:.0f} utt/s", flush=True)
    dt = timeit(lambda: cae(xx)); print(f"cae {prec} full forward (recon+latent): {dt*1e3:.3f} ms -> {B/dt:.0f} utt/s", flush=True)
