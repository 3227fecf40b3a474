import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dfa_amd.model_cae import ConvAutoencoder
dev = torch.device("cuda", 0)
B = 256
g = torch.Generator().manual_seed(1)
x32 = torch.randn(B, 321, 180, generator=g).to(dev)
for prec in ("bf16", "fp32"):
    torch.manual_seed(0)
    m = ConvAutoencoder(precision=prec).to(dev).train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4, weight_decay=1e-4, fused=True)
    x = x32.to(torch.bfloat16) if prec == "bf16" else x32
    def step():
        recon, _ = m(x)
        loss = torch.nn.functional.mse_loss(recon, x32)
        opt.zero_grad(); loss.backward(); opt.step()
        return loss
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): loss = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"cae {prec}: train step {dt*1e3:.2f} ms -> {B/dt:.0f} utt/s  loss {loss.item():.4f}", flush=True)
