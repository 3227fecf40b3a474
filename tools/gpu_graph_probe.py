"""Probe: the CNN2D / CNN1D eval forward captured into a HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm) and replayed, against
eager launches, for small batches where the launches' host cost is comparable to the kernels (latency per call, results equal)."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dfa_amd.model import CNN2D
from dfa_amd.model_cnn1d import CNN1D
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for name, model, dt in (("cnn2d bf16", CNN2D(precision="bf16").to(dev).eval(), torch.bfloat16), ("cnn1d", CNN1D().to(dev).eval(), torch.float32)):
    for B in (1, 4, 16):
        x = torch.randn(B, 180, 321, device=dev).to(dt).transpose(1, 2)
        with torch.no_grad():
            for _ in range(5): ref = model(x)
            torch.cuda.synchronize()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3): model(x)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = model(x)
            g.replay(); torch.cuda.synchronize()
            same = torch.equal(out, ref)
            def t(fn, n=300):
                for _ in range(20): fn()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(n): fn()
                torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
            te, tg = t(lambda: model(x)), t(g.replay)
        print(f"{name} B={B}: eager {te:.1f} us/call, graph replay {tg:.1f} us/call, equal={same}", flush=True)
