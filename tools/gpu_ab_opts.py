"""A/B context options in one process (interleaved rounds).  usage: gpu_ab_opts.py name=v[,name=v...] [more configs]"""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from dfa_amd import _lib
configs = [dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in arg.split(",")) for arg in sys.argv[1:]]
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
x = (torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07).to(device=dev, dtype=torch.bfloat16).transpose(1, 2)
ctx = _lib.Context.get(dev)
model = bench.build_model(torch, dev, "bf16")
ref = None
res = [[] for _ in configs]
for rnd in range(6):
    for ci, cfg in enumerate(configs):
        for k, v in cfg.items(): ctx.set_option(k, v)
        for _ in range(3): out = model(x)
        if ref is None: ref = out.clone()
        err = (out - ref).abs().max().item()
        ctx.timing_reset(); ctx.timing(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): model(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        ctx.timing(False)
        sl = [ctx.timing_read(s) for s in range(4)]
        res[ci].append((dt * 1e3, [round(ms / max(n, 1), 4) for ms, n in sl], err))
for ci, cfg in enumerate(configs):
    ms = sorted(r[0] for r in res[ci])
    k = [sorted(r[1][i] for r in res[ci])[len(ms) // 2] for i in range(4)]
    print(cfg, "step ms median %.4f min %.4f" % (ms[len(ms)//2], ms[0]), "kernel medians", k, "maxdiff vs first", max(r[2] for r in res[ci]), flush=True)
