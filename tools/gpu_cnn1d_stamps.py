"""Phase split of the fused CNN1D kernel (s_memtime stamps at its phase boundaries, context option clock_probe)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfa_amd import _lib
from dfa_amd.model_cnn1d import CNN1D
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)
m = CNN1D().cuda().eval()
g = torch.Generator().manual_seed(1)
x = (torch.randn(B, 180, 321, generator=g) * 3.2).cuda().transpose(1, 2)
ctx = _lib.Context.get(x.device)
for _ in range(200): m(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): m(x)
torch.cuda.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per forward")
ctx.set_option("clock_probe", 1)
for _ in range(20): m(x)
buf = (C.c_longlong * 1024)()
_lib.check(ctx.handle, ctx.lib.dfa_ctx_debug_read(ctx.handle, buf, 1024))
ctx.set_option("clock_probe", 0)
w = np.array(buf[:], dtype=np.int64).reshape(128, 8)[:min(B, 128)]
d = np.stack([w[:, 1] - w[:, 0], w[:, 2] - w[:, 1], w[:, 3] - w[:, 2], w[:, 3] - w[:, 0]], 1)
ghz = (w[:, 3] - w[:, 0]) / ((w[:, 6] - w[:, 5]) * 10.0)
print("cycles (median over workgroups): layer1 %d, layer2 %d, layer3 %d, total %d; clock %.2f GHz -> %.1f us in-kernel" % (
    *np.median(d, 0), np.median(ghz), np.median(d[:, 3]) / np.median(ghz) / 1e3))
print("ideal MFMA cycles: layer1 %d layer2 %d layer3 %d" % (3 * 276 * 64, 6 * 48 * 64, 11 * 96 * 64))
