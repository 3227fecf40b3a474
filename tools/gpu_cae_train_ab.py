"""A/B of the auto-encoder training step (CaeNativeTrainer, [256,321,180] bf16) over one context option: name=v name=v ..."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dfa_amd import _lib
from dfa_amd.model_cae import ConvAutoencoder
from dfa_amd.training.train_step import make_cae_trainer
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
x = torch.randn(256, 321, 180, generator=g).to(dev, torch.bfloat16)
ctx = _lib.Context.get(dev)
arms = [a.split("=") for a in sys.argv[1:]] or [["cae_dgrad_mfma", "0"], ["cae_dgrad_mfma", "1"]]
tr, times, losses = {}, {}, {}
for i, (n, v) in enumerate(arms):
    torch.manual_seed(0)
    tr[i] = make_cae_trainer(ConvAutoencoder(precision="bf16").to(dev), lr=1e-4, weight_decay=1e-4)
    times[i] = []
for rep in range(4):
    for i, (n, v) in enumerate(arms):
        ctx.set_option(n, int(v))
        for _ in range(2): tr[i].step(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): loss = tr[i].step(x)
        torch.cuda.synchronize(); times[i].append((time.perf_counter() - t0) / 8 * 1e3)
        losses[i] = float(loss)
for i, (n, v) in enumerate(arms):
    t = sorted(times[i])
    print(f"{n}={v}: {t[len(t)//2]:.3f} ms per step (min {t[0]:.3f}), loss {losses[i]:.6f}", flush=True)
