"""Distance of the auto-encoder's bf16 training gradients from the rounding-faithful oracle, per statistics mode (cae_conv_stats)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfa_amd import _lib
from dfa_amd.model_cae import ConvAutoencoder
from oracle import torch_ref as R
g = np.load("tests/golden/cae_train.npz")
sd = {k[len("init.sd."):]: v for k, v in g.items() if k.startswith("init.sd.")}
init = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
ctx = _lib.Context.get(torch.device("cuda", 0))
gen = torch.Generator().manual_seed(4)
cases = [torch.randn(24, 96, 180, generator=gen), torch.randn(8, 321, 180, generator=gen) * 2.0]
def gpu_grads(x):
    m = ConvAutoencoder(precision="bf16"); m.load_state_dict(init); m = m.to("cuda").train()
    xb = x.to("cuda").to(torch.bfloat16)
    recon, _ = m(xb)
    torch.nn.MSELoss()(recon.float(), xb.float()).backward()
    return {n: p.grad.float().cpu() for n, p in m.named_parameters()}
for x in cases:
    _, ref = R.cae_train_step_emulated(sd, x, None)
    emu = {k: R.cae_train_step_emulated(sd, x, "bf16", stats=k)[1] for k in ("epilogue", "stored")}
    for arm in (0, 1):
        ctx.set_option("cae_conv_stats", arm)
        got = gpu_grads(x)
        for k in emu:
            out = []
            for n in ("encoder.0.weight", "encoder.4.weight", "encoder.8.weight", "encoder.12.weight", "decoder.0.weight", "decoder.6.weight"):
                scale = max(float(ref[n].abs().max()), 1e-9)
                out.append(f"{n.split('.')[0][:3]}{n.split('.')[1]}={float((got[n] - emu[k][n]).abs().max()) / scale:.3f}")
            print(tuple(x.shape), "gpu cae_conv_stats=%d" % arm, "oracle stats=%s" % k, " ".join(out), flush=True)
    d = []
    for n in ("encoder.0.weight", "encoder.12.weight", "decoder.0.weight"):
        scale = max(float(ref[n].abs().max()), 1e-9)
        d.append(f"{float((emu['epilogue'][n] - emu['stored'][n]).abs().max()) / scale:.3f}")
    print(tuple(x.shape), "oracle epilogue vs oracle stored:", d, flush=True)
ctx.set_option("cae_conv_stats", 1)
