"""A/B two builds of the library (separate processes, alternating): DFA_LIB=<file name under lib/>."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from dfa_amd import _lib
name = os.environ.get("DFA_LIB")
if name:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), name)
import bench
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
x = (torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07).to(device=dev, dtype=torch.bfloat16).transpose(1, 2)
ctx = _lib.Context.get(dev)
model = bench.build_model(torch, dev, "bf16")
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
