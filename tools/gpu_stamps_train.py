"""Diagnostic: run bf16 CNN2D training steps (and the bf16x3 forward) on the stamped build (libdfa_hip_stamps.so, -DDFA_STAMPS): the
conv_split launchers print the per-wave cycle split of an iteration of the data-gradient / bf16x3 kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfa_amd import _lib
_lib.LIB_PATH = _lib.LIB_PATH.replace("libdfa_hip.so", "libdfa_hip_stamps.so")
import bench
from dfa_amd.model import CNN2D
from dfa_amd.training.train_step import NativeTrainer
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
stored = torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07
x = stored.to(dev, dtype=torch.bfloat16).transpose(1, 2)
y = (torch.rand(256, generator=g) > 0.5).float().to(dev)
torch.manual_seed(0)
tr = NativeTrainer(CNN2D(dropout=0.2, precision="bf16").to(dev), label_smoothing=0.05)
for _ in range(35): tr.step(x, y)
torch.cuda.synchronize()
m3 = bench.build_model(torch, dev, "bf16x3")
xf = stored.to(dev).transpose(1, 2)
for _ in range(35): m3(xf)
torch.cuda.synchronize()
