"""A/B of the CNN1D training step (all-C-ABI NativeTrainer, [256,321,180] fp32) over context option cnn1d_train_x3:
0 = fp32 VALU convolutions, 1 = matrix-core layer kernel with three bf16 terms per operand (fp32-grade), 2 = the same with one
channel tile per workgroup, 3 = two terms (bf16x3).
Prints ms per step (median of 5 x 20 steps, arms interleaved) and the loss after the same number of steps."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dfa_amd import _lib
from dfa_amd.model_cnn1d import CNN1D
from dfa_amd.training.train_step import NativeTrainer
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
x = (torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07).to(dev).transpose(1, 2)
y = (torch.rand(256, generator=g) > 0.5).float().to(dev)
ctx = _lib.Context.get(dev)
arms = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3]
trainers, times, losses = {}, {a: [] for a in arms}, {}
for a in arms:
    torch.manual_seed(0)
    trainers[a] = NativeTrainer(CNN1D(dropout=0.2).to(dev), label_smoothing=0.05)
for rep in range(5):
    for a in arms:
        ctx.set_option("cnn1d_train_x3", a)
        tr = trainers[a]
        for _ in range(3): tr.step(x, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): loss = tr.step(x, y)
        torch.cuda.synchronize(); times[a].append((time.perf_counter() - t0) / 20 * 1e3)
        losses[a] = float(loss)
ctx.set_option("cnn1d_train_x3", 1)
for a in arms:
    t = sorted(times[a])
    print(f"cnn1d_train_x3={a}: {t[len(t)//2]:.3f} ms per step (min {t[0]:.3f}), loss after {5*23} steps {losses[a]:.6f}", flush=True)
