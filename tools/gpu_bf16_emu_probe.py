"""Calibration probe (GPU box): bf16-mode logits vs the rounding-faithful oracle on the golden cases; prints the
distances the tolerances in tests/test_parity_r2_gpu.py are set from."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from oracle import dfa_oracle as O
from dfa_amd.model import CNN2D
sd, g = load_golden("cnn2d_eval")
m = CNN2D(precision="bf16"); m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}); m = m.to("cuda").eval()
for tag in ("t321", "t64", "t7"):
    xs = g[f"{tag}.x_stored"]
    got, emb = m(torch.from_numpy(xs).to("cuda").transpose(1, 2), return_embedding=True)
    want, inter = O.cnn2d_forward(sd, np.swapaxes(xs, 1, 2), return_intermediates=True, emulate="bf16")
    ref = g[f"{tag}.logits"]
    print(tag, "logits", want.ravel()[:4], "| vs emulated", np.abs(got.cpu().numpy() - want).max(), "| vs fp32 reference",
          np.abs(got.cpu().numpy() - ref).max(), "| emb vs emulated", np.abs(emb.cpu().numpy() - inter["embedding"]).max(),
          "emb scale", np.abs(inter["embedding"]).max())
