"""Diagnostic: run the CNN2D bf16 forward on the stamped build (libdfa_hip_stamps.so, -DDFA_STAMPS) and let the
launcher print the per-wave cycle split of block 3."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfa_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libdfa_hip.so", "libdfa_hip_stamps.so")
import bench
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1234)
x = (torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07).to(device=dev, dtype=torch.bfloat16).transpose(1, 2)
ctx = _lib.Context.get(dev)
ctx.set_option("conv_dma", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
model = bench.build_model(torch, dev, "bf16")
for _ in range(4100): model(x)
torch.cuda.synchronize()
