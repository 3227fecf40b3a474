"""Plain-PyTorch fp32 CPU restatement of the three forwards -- TEST INFRASTRUCTURE ONLY (same rules as
dfa_oracle.py: imported only by tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py).

It rebuilds the reference's layer stacks from torch.nn.functional calls on a state_dict, which is what the
reference's CPU path executes (src/model.py:33-42, src/model_cnn1d.py:37-46, src/model_cae.py:83-125 under
model.eval()); it is used where the numpy oracle would be too slow (full-size batches, the CPU baseline
timing) and is itself pinned against the same golden vectors (tests/test_oracle_golden.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _t(sd, k):
    v = sd[k]
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def _bn(sd, p, x):
    return F.batch_norm(x, _t(sd, p + ".running_mean"), _t(sd, p + ".running_var"), _t(sd, p + ".weight"),
                        _t(sd, p + ".bias"), training=False, eps=1e-5)


@torch.no_grad()
def cnn2d_forward(sd, x, return_embedding=False):
    h = x.unsqueeze(1)
    h = F.avg_pool2d(F.relu(_bn(sd, "conv.1", F.conv2d(h, _t(sd, "conv.0.weight"), _t(sd, "conv.0.bias"), padding=1))), (2, 1))
    h = F.avg_pool2d(F.relu(_bn(sd, "conv.6", F.conv2d(h, _t(sd, "conv.5.weight"), _t(sd, "conv.5.bias"), padding=1))), (2, 1))
    h = F.relu(_bn(sd, "conv.11", F.conv2d(h, _t(sd, "conv.10.weight"), _t(sd, "conv.10.bias"), padding=1)))
    emb = h.mean(dim=2).flatten(1)
    logits = F.linear(emb, _t(sd, "classifier.weight"), _t(sd, "classifier.bias"))
    return (logits, emb) if return_embedding else logits


@torch.no_grad()
def cnn1d_forward(sd, x):
    h = x.transpose(1, 2)
    h = F.relu(_bn(sd, "conv.1", F.conv1d(h, _t(sd, "conv.0.weight"), _t(sd, "conv.0.bias"), padding=1)))
    h = F.relu(_bn(sd, "conv.5", F.conv1d(h, _t(sd, "conv.4.weight"), _t(sd, "conv.4.bias"), padding=1)))
    h = F.relu(_bn(sd, "conv.9", F.conv1d(h, _t(sd, "conv.8.weight"), _t(sd, "conv.8.bias"), padding=1)))
    return F.linear(h.mean(dim=2), _t(sd, "classifier.weight"), _t(sd, "classifier.bias"))


@torch.no_grad()
def cae_forward(sd, x):
    h = x.unsqueeze(1)
    for c, b in ((0, 1), (4, 5), (8, 9), (12, 13)):
        h = F.conv2d(h, _t(sd, f"encoder.{c}.weight"), _t(sd, f"encoder.{c}.bias"), padding=1)
        h = F.avg_pool2d(F.relu(_bn(sd, f"encoder.{b}", h)), 2)
    latent = h
    d = F.relu(_bn(sd, "decoder.1", F.conv_transpose2d(latent, _t(sd, "decoder.0.weight"), _t(sd, "decoder.0.bias"), stride=2)))
    d = F.relu(_bn(sd, "decoder.4", F.conv_transpose2d(d, _t(sd, "decoder.3.weight"), _t(sd, "decoder.3.bias"), stride=2,
                                                       output_padding=(0, 1))))
    d = F.relu(_bn(sd, "decoder.7", F.conv_transpose2d(d, _t(sd, "decoder.6.weight"), _t(sd, "decoder.6.bias"), stride=2)))
    d = F.conv_transpose2d(d, _t(sd, "decoder.9.weight"), _t(sd, "decoder.9.bias"), stride=2)
    T, Tr = x.size(1), d.size(2)
    if Tr < T:
        d = F.pad(d, (0, 0, 0, T - Tr))
    elif Tr > T:
        d = d[:, :, :T, :]
    return d.squeeze(1), latent


def _bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32)


@torch.no_grad()
def cnn2d_forward_emulated(sd, x, emulate="bf16", return_embedding=False):
    """torch twin of dfa_oracle.cnn2d_forward_emulated (same rounding points of the product's bf16 storage mode, float64
    sums rounded once to fp32), fast enough for full-size batches.  emulate=None = the folded computation without rounding."""
    if emulate not in (None, "bf16"):
        raise ValueError(f"emulate must be None or 'bf16', got {emulate!r}")
    rnd = _bf16r if emulate == "bf16" else (lambda t: t)
    f64 = torch.float64

    def fold(conv, bn):
        s = _t(sd, bn + ".weight").float() / torch.sqrt(_t(sd, bn + ".running_var").float() + 1e-5)
        w = _t(sd, conv + ".weight").float() * s[:, None, None, None]
        b = (_t(sd, conv + ".bias").float() - _t(sd, bn + ".running_mean").float()) * s + _t(sd, bn + ".bias").float()
        return w, b
    (w1, b1), (w2, b2), (w3, b3) = fold("conv.0", "conv.1"), fold("conv.5", "conv.6"), fold("conv.10", "conv.11")
    w1h = 0.5 * w1
    hi = rnd(w1h)
    lo = rnd(w1h - hi)
    w1e, b1e = hi.to(f64) + lo.to(f64), (0.5 * b1)
    w2e, b2e = rnd(w2 * 0.5), b2 * 0.5
    w3e = rnd(w3)

    def conv(h, w, b):
        return F.conv2d(h.to(f64), w.to(f64), b.to(f64), padding=1).float()

    def pooled(z):
        H = z.shape[2] // 2
        z = F.relu(z[:, :, :2 * H])
        return z[:, :, 0::2] + z[:, :, 1::2]
    a1 = rnd(pooled(conv(rnd(x.float()).unsqueeze(1), w1e, b1e)))
    a2 = rnd(pooled(conv(a1, w2e, b2e)))
    a3 = F.relu(conv(a2, w3e, b3))
    emb = (a3.sum(dim=2, dtype=f64).float() * (1.0 / a3.shape[2])).flatten(1)
    logits = F.linear(emb.to(f64), _t(sd, "classifier.weight").to(f64), _t(sd, "classifier.bias").to(f64)).float()
    return (logits, emb) if return_embedding else logits
