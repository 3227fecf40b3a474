"""A/B the conv staging variants in one process (interleaved rounds)."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from dfa_amd import _lib
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
stored = torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07
xs = {"bf16": stored.to(device=dev, dtype=torch.bfloat16).transpose(1, 2), "fp32": stored.to(dev).transpose(1, 2)}
ctx = _lib.Context.get(dev)
for prec in ("bf16", "fp32"):
    model = bench.build_model(torch, dev, prec)
    res = {0: [], 1: []}
    for rnd in range(5):
        for dma in (0, 1):
            ctx.set_option("conv_dma", dma)
            for _ in range(3): model(xs[prec])
            ctx.timing_reset(); ctx.timing(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): model(xs[prec])
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            ctx.timing(False)
            sl = [ctx.timing_read(s) for s in range(4)]
            res[dma].append((dt * 1e3, [round(ms / max(n, 1), 4) for ms, n in sl]))
    for dma in (0, 1):
        ms = sorted(r[0] for r in res[dma])
        print(prec, "dma" if dma else "reg", "step ms median %.4f min %.4f" % (ms[len(ms)//2], ms[0]), "kernels", res[dma][-1][1], flush=True)
