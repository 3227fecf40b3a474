"""A/B two builds of the library (separate processes, alternating): DFA_LIB=<file name under lib/>; DFA_AB_MODE=train times the
bf16 training step instead of the eval forward."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from dfa_amd import _lib
name = os.environ.get("DFA_LIB")
if name:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), name)
import bench
if os.environ.get("DFA_AB_MODE") == "train":
    from dfa_amd.model import CNN2D
    from dfa_amd.training.train_step import NativeTrainer
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07).to(dev, dtype=torch.bfloat16).transpose(1, 2)
    y = (torch.rand(256, generator=g) > 0.5).float().to(dev)
    torch.manual_seed(0)
    tr = NativeTrainer(CNN2D(dropout=0.2, precision="bf16").to(dev), label_smoothing=0.05,
                       lr=float(os.environ.get("DFA_AB_LR", "1e-6")))   # tiny lr: both builds time the same activation regime
    losses = [float(tr.step(x, y)) for _ in range(5)]
    res = []
    for rnd in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): last = tr.step(x, y)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 10 * 1e3)
    print(name, "train step ms median %.3f min %.3f" % (sorted(res)[2], min(res)),
          "losses", ["%.4f" % v for v in losses], "last %.4f" % float(last), flush=True)
    sys.exit(0)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
PREC = os.environ.get("DFA_AB_PREC", "bf16")
x = (torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07).to(device=dev, dtype=torch.bfloat16 if PREC == "bf16" else torch.float32).transpose(1, 2)
ctx = _lib.Context.get(dev)
model = bench.build_model(torch, dev, PREC)
for _ in range(10): model(x)
out = []
for rnd in range(5):
    ctx.timing_reset(); ctx.timing(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): model(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    ctx.timing(False)
    sl = [ctx.timing_read(s) for s in range(4)]
    out.append((round(dt * 1e3, 4), [round(ms / max(n, 1), 4) for ms, n in sl]))
print(name, sorted(out)[len(out)//2], flush=True)
