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


# ---- rounding-faithful training step (bf16 storage mode) -----------------------------------------------------------
class _StoreBF16(torch.autograd.Function):
    """A tensor the product keeps in bf16 in BOTH directions: the forward value is rounded (storage of an activation) and
    so is the gradient that arrives for it (storage of the matching data gradient)."""

    @staticmethod
    def forward(ctx, x):
        return x.float().to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.float().to(torch.bfloat16).to(g.dtype)


class _RoundSTE(torch.autograd.Function):
    """forward: round to bf16; backward: identity (the value is stored rounded, its consumers' gradient passes unchanged)."""

    @staticmethod
    def forward(ctx, x):
        return x.float().to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(torch.autograd.Function):
    """forward: identity; backward: the TOTAL gradient of this node is rounded to bf16 (dz of a BatchNorm backward pass)."""

    @staticmethod
    def forward(ctx, x):
        return x.clone()

    @staticmethod
    def backward(ctx, g):
        return g.float().to(torch.bfloat16).to(g.dtype)


def cnn2d_train_step_emulated(sd, x, y, label_smoothing=0.0, emulate="bf16", round_x=True, return_stats=False):
    """One CNN2D training forward/backward (dropout 0) restated with the ROUNDING POINTS of the product's bf16 training
    mode (src/train.py:71-76 over src/model.py:13-39 in train mode): a1, z2, a2, z3 and their gradients da1, dz2, da2, dz3
    are stored in bf16, the MFMA convolutions of blocks 2 and 3 (forward, data gradient, weight gradient) take bf16
    weights and bf16 activations, BatchNorm batch statistics come from the unrounded accumulators, block 1 / the time mean /
    the classifier / the loss are fp32.  Everything else is float64 with torch autograd.  emulate=None removes every
    rounding: the result must then equal the reference's autograd goldens (pins the restatement).
    round_x=False feeds x unrounded in bf16 mode too (a batch with folded jitter noise reaches block 1 in fp32).
    Returns (logits [B], loss, {parameter name: gradient}) with the names of the model's named_parameters(); with
    return_stats=True a fourth item {BatchNorm prefix: (batch mean, biased batch variance, element count)} from which the
    running-statistics update of nn.BatchNorm2d follows (momentum m: (1-m)*old + m*mean, (1-m)*old + m*var*n/(n-1))."""
    f64 = torch.float64
    on = emulate == "bf16"
    stats = {}
    if emulate not in (None, "bf16"):
        raise ValueError(emulate)
    P = {k: _t(sd, k).to(f64).clone().requires_grad_(True) for k in sd
         if k.endswith(("weight", "bias")) and not k.endswith(("running_mean", "running_var"))}
    store = _StoreBF16.apply if on else (lambda t: t)
    rnd = _RoundSTE.apply if on else (lambda t: t)
    rgrad = _RoundGrad.apply if on else (lambda t: t)

    def bn_train(z, pfx):       # batch statistics of the accumulators, normalisation of the stored value
        za = rgrad(z)
        mean = za.mean(dim=(0, 2, 3), keepdim=True)
        var = za.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        stats[pfx] = (mean.detach().flatten().float(), var.detach().flatten().float(), za.numel() // za.shape[1])
        zs = rnd(za)
        return (zs - mean) / torch.sqrt(var + 1e-5) * P[pfx + ".weight"][None, :, None, None] + P[pfx + ".bias"][None, :, None, None]

    xb = (x.to(torch.bfloat16) if (on and round_x) else x).to(f64).unsqueeze(1)
    # block 1: fp32 VALU convolution from x, z1 is never stored (no rounding of z1 or dz1)
    z1 = F.conv2d(xb, P["conv.0.weight"], P["conv.0.bias"], padding=1)
    m1 = z1.mean(dim=(0, 2, 3), keepdim=True)
    v1 = z1.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    stats["conv.1"] = (m1.detach().flatten().float(), v1.detach().flatten().float(), z1.numel() // z1.shape[1])
    y1 = (z1 - m1) / torch.sqrt(v1 + 1e-5) * P["conv.1.weight"][None, :, None, None] + P["conv.1.bias"][None, :, None, None]
    a1 = store(F.avg_pool2d(F.relu(y1), (2, 1)))
    w2 = P["conv.5.weight"]
    w2b = w2 + (rnd(w2) - w2).detach() if on else w2
    a2 = store(F.avg_pool2d(F.relu(bn_train(F.conv2d(a1, w2b, P["conv.5.bias"], padding=1), "conv.6")), (2, 1)))
    w3 = P["conv.10.weight"]
    w3b = w3 + (rnd(w3) - w3).detach() if on else w3
    a3 = F.relu(bn_train(F.conv2d(a2, w3b, P["conv.10.bias"], padding=1), "conv.11"))
    emb = a3.mean(dim=2).flatten(1)
    logits = F.linear(emb, P["classifier.weight"], P["classifier.bias"]).squeeze(-1)
    ys = y.to(f64) * (1.0 - label_smoothing) + 0.5 * label_smoothing if label_smoothing > 0 else y.to(f64)
    loss = F.binary_cross_entropy_with_logits(logits, ys)
    loss.backward()
    out = (logits.detach().float(), float(loss), {k: v.grad.float() for k, v in P.items()})
    return out + (stats,) if return_stats else out


def cnn1d_train_step(sd, x, y, label_smoothing=0.0, return_stats=False):
    """One CNN1D training forward/backward (dropout 0; src/train.py:71-76 over src/model_cnn1d.py:13-46 in train mode) in float64
    with torch autograd -- the product's CNN1D path is fp32 end to end, so there are no rounding points to restate.  x is the
    [B,T,F] view the harness feeds.  Returns (logits [B], loss, {parameter: gradient}[, {BatchNorm prefix: (mean, biased var, n)}])."""
    f64 = torch.float64
    P = {k: _t(sd, k).to(f64).clone().requires_grad_(True) for k in sd
         if k.endswith(("weight", "bias")) and not k.endswith(("running_mean", "running_var"))}
    stats = {}

    def block(h, conv, bn):
        z = F.conv1d(h, P[conv + ".weight"], P[conv + ".bias"], padding=1)
        mean = z.mean(dim=(0, 2), keepdim=True)
        var = z.var(dim=(0, 2), unbiased=False, keepdim=True)
        stats[bn] = (mean.detach().flatten().float(), var.detach().flatten().float(), z.numel() // z.shape[1])
        return F.relu((z - mean) / torch.sqrt(var + 1e-5) * P[bn + ".weight"][None, :, None] + P[bn + ".bias"][None, :, None])
    h = block(x.to(f64).transpose(1, 2), "conv.0", "conv.1")
    h = block(h, "conv.4", "conv.5")
    h = block(h, "conv.8", "conv.9")
    logits = F.linear(h.mean(dim=2), P["classifier.weight"], P["classifier.bias"]).squeeze(-1)
    ys = y.to(f64) * (1.0 - label_smoothing) + 0.5 * label_smoothing if label_smoothing > 0 else y.to(f64)
    loss = F.binary_cross_entropy_with_logits(logits, ys)
    loss.backward()
    out = (logits.detach().float(), float(loss), {k: v.grad.float() for k, v in P.items()})
    return out + (stats,) if return_stats else out


def state_after_adamw_step(sd, grads, stats, lr=1e-3, weight_decay=0.01, momentum=0.1):
    """The state_dict one optimiser step later: torch.optim.AdamW(lr, weight_decay) on the given gradients (src/train.py:326-328,
    74-76) plus the running-statistics update nn.BatchNorm performs in the train-mode forward (momentum 0.1, unbiased variance).
    `stats` = {BatchNorm prefix: (batch mean, biased batch variance, element count)} as the *_train_step oracles return them."""
    out = {k: _t(sd, k).clone() for k in sd}
    params = [torch.nn.Parameter(out[k].float().clone()) for k in grads]
    for p, k in zip(params, grads):
        p.grad = grads[k].float().clone()
    torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay).step()
    for p, k in zip(params, grads):
        out[k] = p.detach().clone()
    for bn, (mean, var, n) in stats.items():
        out[bn + ".running_mean"] = (1 - momentum) * out[bn + ".running_mean"].float() + momentum * mean
        out[bn + ".running_var"] = (1 - momentum) * out[bn + ".running_var"].float() + momentum * var * (n / max(n - 1, 1))
        if bn + ".num_batches_tracked" in out:
            out[bn + ".num_batches_tracked"] = out[bn + ".num_batches_tracked"] + 1
    return out


def cae_train_step_emulated(sd, x, emulate="bf16", stats="epilogue"):
    """One ConvAutoencoder training forward/backward (src/train_cae.py:58-82 over src/model_cae.py:32-125 in train mode, loss =
    MSELoss(recon, x)) restated with the ROUNDING POINTS of the product's bf16 training mode (csrc/cae_train_api.hip):
      encoder block 1   fp32 convolution from the bf16 features, z1 never stored (statistics of the unrounded z1); its pooled
                        output e1 and the gradient de1 that arrives for it are stored in bf16;
      encoder 2-4       MFMA convolutions with bf16 weights on bf16 inputs; the pre-BatchNorm output z is STORED in bf16; the
                        batch statistics of blocks 2 and 3 are the convolution epilogue's fp32 sums of the UNROUNDED outputs
                        (context option cae_conv_stats = 1, the default; stats="stored" restates the separate pass over the
                        stored tensor of option 0 and of round 2 -- which block 4 keeps), the normalisation reads the stored tensor; the BatchNorm backward writes dz in
                        bf16; pooled outputs e and their gradients de in bf16;
      decoder 1-3       ConvTranspose2d(k2, s2) with bf16 weights, zd / dzd / d / dd stored in bf16, statistics as in the encoder;
      decoder 4         fp32 from the bf16 d3.
    (The product's ConvTranspose2d DATA gradient multiplies by the unrounded fp32 weights; here the rounded ones are used in both
    directions -- a 2^-9 relative difference per weight, below the bf16 storage of the result.)  Everything else is float64
    autograd.  emulate=None removes every rounding: the result must equal the reference's autograd goldens.
    Returns (loss, {parameter name: gradient})."""
    f64 = torch.float64
    on = emulate == "bf16"
    if emulate not in (None, "bf16"):
        raise ValueError(emulate)
    if stats not in ("epilogue", "stored"):
        raise ValueError(stats)
    P = {k: _t(sd, k).to(f64).clone().requires_grad_(True) for k in sd
         if k.endswith(("weight", "bias")) and not k.endswith(("running_mean", "running_var"))}
    store = _StoreBF16.apply if on else (lambda t: t)
    rnd = _RoundSTE.apply if on else (lambda t: t)
    rgrad = _RoundGrad.apply if on else (lambda t: t)

    def wq(name):               # bf16 weight image of an MFMA layer (gradient passes to the fp32 parameter)
        w = P[name]
        return w + (rnd(w) - w).detach() if on else w

    def bn(z, pfx, zu=None):
        # zu: the unrounded values the statistics are summed from (their gradient still arrives through the stored tensor's dz)
        zs = z if (zu is None or stats == "stored") else z + (zu - z).detach()
        mean = zs.mean(dim=(0, 2, 3), keepdim=True)
        var = zs.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        return (z - mean) / torch.sqrt(var + 1e-5) * P[pfx + ".weight"][None, :, None, None] + P[pfx + ".bias"][None, :, None, None]

    def stored(z):              # pre-BN output kept in bf16 (forward value rounded), dz in bf16
        return rnd(rgrad(z))

    xb = (x.to(torch.bfloat16) if on else x).to(f64)
    h = xb.unsqueeze(1)
    h = store(F.avg_pool2d(F.relu(bn(F.conv2d(h, P["encoder.0.weight"], P["encoder.0.bias"], padding=1), "encoder.1")), 2))
    for c, b in ((4, 5), (8, 9), (12, 13)):
        zu = F.conv2d(h, wq(f"encoder.{c}.weight"), P[f"encoder.{c}.bias"], padding=1)
        h = store(F.avg_pool2d(F.relu(bn(stored(zu), f"encoder.{b}", zu if c != 12 else None)), 2))
    d = h
    for c, b, opad in ((0, 1, (0, 0)), (3, 4, (0, 1)), (6, 7, (0, 0))):
        zu = F.conv_transpose2d(d, wq(f"decoder.{c}.weight"), P[f"decoder.{c}.bias"], stride=2, output_padding=opad)
        d = store(F.relu(bn(stored(zu), f"decoder.{b}", zu)))
    r = F.conv_transpose2d(d, P["decoder.9.weight"], P["decoder.9.bias"], stride=2)
    T, Tr = x.size(1), r.size(2)
    if Tr < T:
        r = F.pad(r, (0, 0, 0, T - Tr))
    elif Tr > T:
        r = r[:, :, :T, :]
    loss = F.mse_loss(r.squeeze(1), xb)
    loss.backward()
    return float(loss.detach()), {k: v.grad.float() for k, v in P.items()}


@torch.no_grad()
def cae_forward_emulated(sd, x, emulate="bf16"):
    """ConvAutoencoder.forward (eval, src/model_cae.py:83-125) restated with the ROUNDING POINTS of the product's bf16 mode:
    encoder block 1 in fp32 (folded weights, 1/4 pool factor after the ReLUs) with its output stored in bf16; encoder
    blocks 2-4 and decoder blocks 1-3 with bf16 weights (BatchNorm folded; the encoder's 1/4 pool factor folded into
    weights and bias before the rounding), fp32 bias, float64 sums rounded once, outputs stored in bf16 (the latent map is
    the bf16 value); decoder block 4 in fp32 from the bf16 d3.  emulate=None: no rounding (must equal the reference).
    Returns (reconstruction [B,T,F], latent [B,256,T/16,F/16])."""
    if emulate not in (None, "bf16"):
        raise ValueError(emulate)
    rnd = _bf16r if emulate == "bf16" else (lambda t: t)
    f64 = torch.float64

    def fold(conv, bn, transposed=False):
        s = _t(sd, bn + ".weight").float() / torch.sqrt(_t(sd, bn + ".running_var").float() + 1e-5)
        w = _t(sd, conv + ".weight").float()
        w = w * (s[None, :, None, None] if transposed else s[:, None, None, None])
        b = (_t(sd, conv + ".bias").float() - _t(sd, bn + ".running_mean").float()) * s + _t(sd, bn + ".bias").float()
        return w, b
    h = x.float().unsqueeze(1)
    w, b = fold("encoder.0", "encoder.1")
    z = F.relu(F.conv2d(h.to(f64), w.to(f64), b.to(f64), padding=1).float())
    h = rnd(F.avg_pool2d(z, 2))                                         # 0.25 * sum of four ReLUs in fp32
    for c, bnk in ((4, 5), (8, 9), (12, 13)):
        w, b = fold(f"encoder.{c}", f"encoder.{bnk}")
        wq, bq = rnd(w * 0.25), b * 0.25
        z = F.relu(F.conv2d(h.to(f64), wq.to(f64), bq.to(f64), padding=1).float())
        H2, W2 = z.shape[2] // 2, z.shape[3] // 2
        z = z[:, :, :2 * H2, :2 * W2]
        h = rnd((z[:, :, 0::2, 0::2] + z[:, :, 1::2, 0::2]) + (z[:, :, 0::2, 1::2] + z[:, :, 1::2, 1::2]))
    latent = h
    d = latent
    for c, bnk, opad in ((0, 1, (0, 0)), (3, 4, (0, 1)), (6, 7, (0, 0))):
        w, b = fold(f"decoder.{c}", f"decoder.{bnk}", transposed=True)
        d = rnd(F.relu(F.conv_transpose2d(d.to(f64), rnd(w).to(f64), b.to(f64), stride=2, output_padding=opad).float()))
    d = F.conv_transpose2d(d.to(f64), _t(sd, "decoder.9.weight").to(f64), _t(sd, "decoder.9.bias").to(f64), stride=2).float()
    T, Tr = x.size(1), d.size(2)
    if Tr < T:
        d = F.pad(d, (0, 0, 0, T - Tr))
    elif Tr > T:
        d = d[:, :, :T, :]
    return d.squeeze(1), latent
