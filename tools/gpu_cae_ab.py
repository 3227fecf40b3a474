"""Interleaved A/B of context options on the auto-encoder score at [256,321,180] bf16: tools/gpu_cae_ab.py cae_enc_dma=0 cae_enc_dma=1 ..."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfa_amd import _lib
from dfa_amd.model_cae import ConvAutoencoder
torch.manual_seed(0)
m = ConvAutoencoder(precision="bf16").cuda().eval()
g = torch.Generator().manual_seed(1)
x = torch.randn(256, 180, 321, generator=g).cuda().to(torch.bfloat16).transpose(1, 2)
ctx = _lib.Context.get(x.device)
arms = [a.split("=") for a in sys.argv[1:]] or [["cae_enc_dma", "0"], ["cae_enc_dma", "1"]]
ref = None
res = {tuple(a): [] for a in arms}
for rep in range(5):
    for name, val in arms:
        ctx.set_option(name, int(val))
        for _ in range(20): s = m.score(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50): s = m.score(x)
        torch.cuda.synchronize()
        res[(name, val)].append((time.perf_counter() - t0) / 50 * 1e3)
        if ref is None: ref = s.clone()
        print(name, val, "max |score - first arm|", float((s - ref).abs().max()), "bit-equal" if torch.equal(s, ref) else "")
for k, v in res.items():
    print(k, "ms per score: min %.4f median %.4f" % (min(v), sorted(v)[len(v) // 2]))
ctx.timing_reset(); 
for name, val in arms:
    ctx.set_option(name, int(val)); ctx.timing_reset(); ctx.timing(True)
    for _ in range(20): m.score(x)
    ctx.timing(False); torch.cuda.synchronize()
    print(name, val, {n: round(ctx.timing_read(sl)[0] / max(ctx.timing_read(sl)[1], 1), 4) for n, sl in (("enc1", 8), ("enc2", 9), ("enc3", 10), ("enc4", 11), ("dec", 12))})
