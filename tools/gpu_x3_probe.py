"""bf16x3 probe (GPU box): distance from the reference goldens and a first timing of the three launches."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from dfa_amd.model import CNN2D
from dfa_amd import _lib
sd, g = load_golden("cnn2d_eval")
def mk(prec):
    m = CNN2D(precision=prec); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); return m.to("cuda").eval()
m = mk("bf16x3"); m32 = mk("fp32")
for tag in ("t321", "t64", "t7", "t16"):
    x = torch.from_numpy(g[f"{tag}.x_stored"]).to("cuda").transpose(1, 2)
    got = m(x).cpu().numpy(); ref = g[f"{tag}.logits"]; f32 = m32(x).cpu().numpy()
    print(tag, "x3 vs reference", np.abs(got - ref).max(), "| fp32 mode vs reference", np.abs(f32 - ref).max(), flush=True)
gen = torch.Generator().manual_seed(1)
x = (torch.randn(256, 180, 321, generator=gen) * 3.2 - 0.07).to("cuda").transpose(1, 2)
ctx = _lib.Context.get(x.device)
for _ in range(10): m(x)
torch.cuda.synchronize()
ctx.timing_reset(); ctx.timing(True)
t0 = time.perf_counter()
for _ in range(30): m(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 30
ctx.timing(False)
print("bf16x3 B=256: %.3f ms/step -> %.0f utt/s" % (dt * 1e3, 256 / dt), [round(ctx.timing_read(s)[0] / max(ctx.timing_read(s)[1], 1), 4) for s in range(4)], flush=True)
