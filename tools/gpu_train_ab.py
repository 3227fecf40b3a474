"""A/B the bf16 training-step kernel variants in one process: gpu_train_ab.py train_conv_variant=1 train_conv_variant=2 ..."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dfa_amd import _lib
from dfa_amd.model import CNN2D
from dfa_amd.training.train_step import NativeTrainer
configs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in arg.split(",")) for arg in sys.argv[1:]]
dev = torch.device("cuda", 0)
ctx = _lib.Context.get(dev)
B = 256
g = torch.Generator().manual_seed(1)
x = (torch.randn(B, 180, 321, generator=g) * 3.2 - 0.07).to(dev, dtype=torch.bfloat16).transpose(1, 2)
y = (torch.rand(B, generator=g) > 0.5).float().to(dev)
torch.manual_seed(0)
tr = NativeTrainer(CNN2D(dropout=0.2, precision="bf16").to(dev), label_smoothing=0.05)
res = [[] for _ in configs]
for rnd in range(5):
    for ci, cfg in enumerate(configs):
        for k, v in cfg.items(): ctx.set_option(k, v)
        for _ in range(2): tr.step(x, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): tr.step(x, y)
        torch.cuda.synchronize(); res[ci].append((time.perf_counter() - t0) / 8 * 1e3)
for ci, cfg in enumerate(configs):
    ms = sorted(res[ci]); print(cfg, "train step ms median %.3f min %.3f" % (ms[len(ms) // 2], ms[0]), flush=True)
