"""Phase split of the fused auto-encoder decoder kernel (s_memtime stamps, context option clock_probe)."""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfa_amd import _lib
from dfa_amd.model_cae import ConvAutoencoder
B = 256
torch.manual_seed(0)
m = ConvAutoencoder(precision="bf16").cuda().eval()
g = torch.Generator().manual_seed(1)
layout = sys.argv[1] if len(sys.argv) > 1 else "bft"
x = (torch.randn(B, 180, 321, generator=g)).cuda().to(torch.bfloat16).transpose(1, 2) if layout == "bft" else torch.randn(B, 321, 180, generator=g).cuda().to(torch.bfloat16)
ctx = _lib.Context.get(x.device)
for _ in range(30): m.score(x)
ctx.set_option("clock_probe", 1)
for _ in range(5): m.score(x)
buf = (C.c_longlong * 1024)()
_lib.check(ctx.handle, ctx.lib.dfa_ctx_debug_read(ctx.handle, buf, 1024))
ctx.set_option("clock_probe", 0)
w = np.array(buf[:], dtype=np.int64).reshape(128, 8)
d = np.stack([w[:, 1] - w[:, 0], w[:, 2] - w[:, 1], w[:, 3] - w[:, 2], w[:, 4] - w[:, 3], w[:, 4] - w[:, 0]], 1)
ghz = (w[:, 4] - w[:, 0]) / ((w[:, 7] - w[:, 6]) * 10.0)
print(layout, "cycles (median): stage %d, phase A %d, phase B %d, phase C %d, total %d; clock %.2f GHz" % (*np.median(d, 0), np.median(ghz)))
