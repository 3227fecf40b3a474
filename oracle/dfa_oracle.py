"""CPU oracle for the LFCC-classifier hot path -- TEST INFRASTRUCTURE ONLY.

A plain-numpy restatement of the reference's algorithm for the path named by
BASELINE.json `north_star` (CNN2D / CNN1D / ConvAutoencoder forward, the losses,
the AdamW step, the EER scorer, the CAE normaliser and the score fusion).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import this module, and only as the checker.  The product path
(`dfa_amd` -> ctypes -> libdfa_hip.so) never imports it and has no CPU fallback.

Pinned: every function here is checked against golden vectors produced by
importing the reference's own modules (tests/golden/make_golden.py, run in the build
container where /root/reference is mounted); see tests/test_oracle_golden.py.

All arithmetic is fp32 unless `dtype=np.float64` is passed (used as a higher-precision
yardstick when judging which of two fp32 results is closer to the truth).
Citations are `path:line` relative to the reference repository root.
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5  # torch.nn.BatchNorm{1,2}d default, src/model.py:16


# --------------------------------------------------------------------------- layers
def conv2d_3x3(x, w, b, dtype=np.float32):
    """Conv2d(k=3, padding=1, stride=1) -- src/model.py:15,21,27; src/model_cae.py:34-52.

    x [B,Ci,H,W], w [Co,Ci,3,3], b [Co] -> [B,Co,H,W].  Cross-correlation (torch semantics):
    out[b,co,t,f] = b[co] + sum_{ci,dy,dx} w[co,ci,dy,dx] * xpad[b,ci,t+dy,f+dx].
    """
    x = np.asarray(x, dtype)
    w = np.asarray(w, dtype)
    B, Ci, H, W = x.shape
    Co = w.shape[0]
    xp = np.zeros((B, Ci, H + 2, W + 2), dtype)
    xp[:, :, 1:-1, 1:-1] = x
    out = np.zeros((B, Co, H, W), dtype)
    for dy in range(3):
        for dx in range(3):
            win = xp[:, :, dy:dy + H, dx:dx + W]                      # [B,Ci,H,W]
            out += np.einsum("oc,bchw->bohw", w[:, :, dy, dx], win, optimize=True).astype(dtype)
    out += np.asarray(b, dtype)[None, :, None, None]
    return out


def conv1d_k3(x, w, b, dtype=np.float32):
    """Conv1d(k=3, padding=1) -- src/model_cnn1d.py:17,23,29.  x [B,Ci,T], w [Co,Ci,3]."""
    x = np.asarray(x, dtype)
    w = np.asarray(w, dtype)
    B, Ci, T = x.shape
    xp = np.zeros((B, Ci, T + 2), dtype)
    xp[:, :, 1:-1] = x
    out = np.zeros((B, w.shape[0], T), dtype)
    for k in range(3):
        out += np.einsum("oc,bct->bot", w[:, :, k], xp[:, :, k:k + T], optimize=True).astype(dtype)
    out += np.asarray(b, dtype)[None, :, None]
    return out


def conv_transpose2d_k2s2(x, w, b, output_padding=(0, 0), dtype=np.float32):
    """ConvTranspose2d(k=2, stride=2[, output_padding]) -- src/model_cae.py:63,68-69,74,79.

    w layout is (Cin, Cout, 2, 2).  k == stride => non-overlapping:
    out[b,co,2i+a,2j+c] = b[co] + sum_ci x[b,ci,i,j] * w[ci,co,a,c];
    rows/cols added by output_padding receive the bias only.
    """
    x = np.asarray(x, dtype)
    w = np.asarray(w, dtype)
    B, Ci, H, W = x.shape
    Co = w.shape[1]
    Ho, Wo = 2 * H + output_padding[0], 2 * W + output_padding[1]
    out = np.zeros((B, Co, Ho, Wo), dtype)
    y = np.einsum("bchw,coak->bohawk", x, w, optimize=True).astype(dtype)  # [B,Co,H,2,W,2]
    out[:, :, :2 * H, :2 * W] = y.reshape(B, Co, 2 * H, 2 * W)
    out += np.asarray(b, dtype)[None, :, None, None]
    return out


def batchnorm_eval(x, gamma, beta, mean, var, eps=BN_EPS, dtype=np.float32):
    """BatchNorm in eval mode (running stats) -- src/model.py:16,22,28 under model.eval()
    (src/predict.py:87).  Channel axis is 1."""
    x = np.asarray(x, dtype)
    shp = [1, -1] + [1] * (x.ndim - 2)
    inv = (1.0 / np.sqrt(np.asarray(var, dtype) + dtype(eps))).astype(dtype)
    return ((x - np.asarray(mean, dtype).reshape(shp)) * inv.reshape(shp)
            * np.asarray(gamma, dtype).reshape(shp) + np.asarray(beta, dtype).reshape(shp)).astype(dtype)


def batchnorm_train(x, gamma, beta, eps=BN_EPS, dtype=np.float32):
    """BatchNorm in train mode: normalise with the *biased* batch variance; returns
    (y, batch_mean, biased_var, unbiased_var) so callers can restate the running-stat update
    running = (1-m)*running + m*stat (momentum m=0.1, unbiased var) of torch.nn.BatchNorm."""
    x = np.asarray(x, dtype)
    axes = tuple(i for i in range(x.ndim) if i != 1)
    n = x.size // x.shape[1]
    mean = x.mean(axis=axes, dtype=np.float64)
    var_b = x.var(axis=axes, dtype=np.float64)
    var_u = var_b * n / max(n - 1, 1)
    shp = [1, -1] + [1] * (x.ndim - 2)
    inv = 1.0 / np.sqrt(var_b + eps)
    y = (x - mean.reshape(shp).astype(dtype)) * inv.reshape(shp).astype(dtype)
    y = y * np.asarray(gamma, dtype).reshape(shp) + np.asarray(beta, dtype).reshape(shp)
    return y.astype(dtype), mean.astype(dtype), var_b.astype(dtype), var_u.astype(dtype)


def relu(x):
    return np.maximum(x, 0)


def avgpool2d(x, kh, kw):
    """AvgPool2d(kernel=(kh,kw)), stride = kernel, floor mode -- src/model.py:18,24 ((2,1));
    src/model_cae.py:37,43,49,55 ((2,2)).  Trailing rows/cols that do not fill a window are dropped."""
    B, C, H, W = x.shape
    Ho, Wo = H // kh, W // kw
    v = x[:, :, :Ho * kh, :Wo * kw].reshape(B, C, Ho, kh, Wo, kw)
    return (v.sum(axis=(3, 5), dtype=x.dtype) * x.dtype.type(1.0 / (kh * kw))).astype(x.dtype)


def linear(x, w, b, dtype=np.float32):
    """nn.Linear: x [B,K] @ w[N,K]^T + b -- src/model.py:31,39; src/model_cnn1d.py:35,45."""
    return (np.asarray(x, dtype) @ np.asarray(w, dtype).T + np.asarray(b, dtype)).astype(dtype)


def sigmoid(z):
    z = np.asarray(z, np.float64)
    return 1.0 / (1.0 + np.exp(-z))


# --------------------------------------------------------------------------- models
def _bn(sd, prefix, x, dtype):
    return batchnorm_eval(x, sd[prefix + ".weight"], sd[prefix + ".bias"],
                          sd[prefix + ".running_mean"], sd[prefix + ".running_var"], dtype=dtype)


def bf16_round(a):
    """float32 -> bfloat16 (round to nearest even, the rounding of v_cvt_pk_bf16_f32 and torch's .to(bfloat16)) ->
    float32.  Finite inputs only (the oracle never sees NaN/inf)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000))
    return r.view(np.float32).reshape(np.shape(a))


def _fold_bn(sd, conv, bn):
    """eval-mode BatchNorm folded into the preceding convolution, in fp32 and in the operation order of the product's
    weight preparation: s = gamma / sqrt(var + eps); w' = w * s; b' = (b - mean) * s + beta  (src/model.py:15-16 etc.)."""
    f = np.float32
    s = (np.asarray(sd[bn + ".weight"], f) / np.sqrt(np.asarray(sd[bn + ".running_var"], f) + f(BN_EPS))).astype(f)
    w = (np.asarray(sd[conv + ".weight"], f) * s[:, None, None, None]).astype(f)
    b = ((np.asarray(sd[conv + ".bias"], f) - np.asarray(sd[bn + ".running_mean"], f)) * s + np.asarray(sd[bn + ".bias"], f)).astype(f)
    return w, b


def cnn2d_forward_emulated(sd, x, emulate="bf16", return_intermediates=False):
    """CNN2D.forward (eval) restated with the ROUNDING POINTS of the product's bf16 storage mode (DFA_PREC_BF16, the
    headline configuration), so the GPU result can be held to accumulation-order noise instead of a loose bf16 bound:

      x -> bf16;  block 1: folded fp32 weights * 1/2 (pool factor) split into bf16 hi + lo, fp32 accumulate, ReLU, pool
      add in fp32, a1 -> bf16;  block 2: (folded weights * 1/2) -> bf16, bias fp32, fp32 accumulate, ReLU, pool add,
      a2 -> bf16;  block 3: folded weights -> bf16, fp32 accumulate, ReLU, mean over T and the classifier in fp32.

    The sums are taken in float64 and rounded once to fp32 (the order-free limit of the kernels' fp32 accumulation).
    emulate=None runs the same folded computation with every rounding replaced by the identity: that form must equal the
    plain restatement `cnn2d_forward` (and the reference's goldens) to fp32 noise -- it pins the folding algebra.
    Follows src/model.py:33-42 (stack :13-31); the rounding points are this project's, not the reference's."""
    if emulate not in (None, "bf16"):
        raise ValueError(f"emulate must be None or 'bf16', got {emulate!r}")
    f, d = np.float32, np.float64
    rnd = bf16_round if emulate == "bf16" else (lambda a: np.asarray(a, f))
    w1, b1 = _fold_bn(sd, "conv.0", "conv.1")
    w2, b2 = _fold_bn(sd, "conv.5", "conv.6")
    w3, b3 = _fold_bn(sd, "conv.10", "conv.11")
    half = f(0.5)
    w1h = (half * w1).astype(f)
    hi = rnd(w1h)
    lo = rnd((w1h - hi).astype(f))
    w1e = hi.astype(d) + lo.astype(d)                       # two exact bf16 x bf16 products per tap in the kernel
    b1e = (half * b1).astype(f)
    w2e = rnd((w2 * half).astype(f))
    b2e = (b2 * half).astype(f)
    w3e = rnd(w3)
    xb = rnd(np.asarray(x, f))[:, None, :, :]

    def pooled(z):                                          # relu(row 2q) + relu(row 2q+1) in fp32; 1/2 is in the weights
        H = z.shape[2] // 2
        z = relu(z[:, :, :2 * H]).astype(f)
        return (z[:, :, 0::2] + z[:, :, 1::2]).astype(f)
    a1 = rnd(pooled(conv2d_3x3(xb, w1e, b1e, d).astype(f)))
    a2 = rnd(pooled(conv2d_3x3(a1, w2e, b2e, d).astype(f)))
    a3 = relu(conv2d_3x3(a2, w3e, b3, d).astype(f))
    H2 = a3.shape[2]
    emb = (a3.sum(axis=2, dtype=d).astype(f) * f(1.0 / H2)).astype(f).reshape(a3.shape[0], -1)
    logits = linear(emb, sd["classifier.weight"], sd["classifier.bias"], d).astype(f)
    if return_intermediates:
        return logits, {"a1": a1, "a2": a2, "a3": a3, "embedding": emb}
    return logits


def cnn2d_forward(sd, x, return_intermediates=False, dtype=np.float32, emulate=None):
    """CNN2D.forward in eval mode -- src/model.py:33-42 with the layer stack of :13-31.

    sd: state_dict as {key: ndarray} (keys conv.{0,1,5,6,10,11}.*, classifier.*).
    x:  [B,T,F] (any strides).  Returns logits [B,1] (and a dict of intermediates).
    emulate="bf16": the same forward with the product's bf16-mode rounding points (cnn2d_forward_emulated).
    """
    if emulate is not None:
        return cnn2d_forward_emulated(sd, x, emulate, return_intermediates)
    h = np.asarray(x, dtype)[:, None, :, :]                                   # :34 unsqueeze(1)
    h = conv2d_3x3(h, sd["conv.0.weight"], sd["conv.0.bias"], dtype)           # :15
    h = relu(_bn(sd, "conv.1", h, dtype))                                      # :16-17
    a1 = avgpool2d(h, 2, 1)                                                    # :18 (dropout :19 = id in eval)
    h = conv2d_3x3(a1, sd["conv.5.weight"], sd["conv.5.bias"], dtype)          # :21
    h = relu(_bn(sd, "conv.6", h, dtype))                                      # :22-23
    a2 = avgpool2d(h, 2, 1)                                                    # :24
    h = conv2d_3x3(a2, sd["conv.10.weight"], sd["conv.10.bias"], dtype)        # :27
    a3 = relu(_bn(sd, "conv.11", h, dtype))                                    # :28-29
    emb = a3.mean(axis=2, dtype=dtype).reshape(a3.shape[0], -1)                # :37-38, order c*F+f
    logits = linear(emb, sd["classifier.weight"], sd["classifier.bias"], dtype)  # :39
    if return_intermediates:
        return logits, {"a1": a1, "a2": a2, "a3": a3, "embedding": emb}
    return logits


def cnn1d_forward(sd, x, return_intermediates=False, dtype=np.float32):
    """CNN1D.forward in eval mode -- src/model_cnn1d.py:37-46 (stack :15-35).  x [B,T,F]."""
    h = np.ascontiguousarray(np.swapaxes(np.asarray(x, dtype), 1, 2))          # :40 -> [B,F,T]
    h = relu(_bn(sd, "conv.1", conv1d_k3(h, sd["conv.0.weight"], sd["conv.0.bias"], dtype), dtype))
    h1 = h
    h = relu(_bn(sd, "conv.5", conv1d_k3(h, sd["conv.4.weight"], sd["conv.4.bias"], dtype), dtype))
    h2 = h
    h = relu(_bn(sd, "conv.9", conv1d_k3(h, sd["conv.8.weight"], sd["conv.8.bias"], dtype), dtype))
    pooled = h.mean(axis=2, dtype=dtype)                                       # :34,42-43
    logits = linear(pooled, sd["classifier.weight"], sd["classifier.bias"], dtype)
    if return_intermediates:
        return logits, {"h1": h1, "h2": h2, "h3": h, "pooled": pooled}
    return logits


def cae_forward(sd, x, return_intermediates=False, dtype=np.float32):
    """ConvAutoencoder.forward in eval mode -- src/model_cae.py:83-125 (stacks :32-80).

    x [B,T,F] already z-scored.  Returns (reconstruction [B,T,F], latent [B,8bc,T/16,F/16]).
    The decoder yields 16*floor(T/16) rows; missing rows are zero-padded, extra rows trimmed
    (:113-121) -- so for T=321 row 320 of the reconstruction is exactly 0.
    """
    inter = {}
    h = np.asarray(x, dtype)[:, None, :, :]
    for i, (c, bn) in enumerate(((0, 1), (4, 5), (8, 9), (12, 13))):           # :34-55
        h = conv2d_3x3(h, sd[f"encoder.{c}.weight"], sd[f"encoder.{c}.bias"], dtype)
        h = avgpool2d(relu(_bn(sd, f"encoder.{bn}", h, dtype)), 2, 2)
        inter[f"enc{i + 1}"] = h
    latent = h
    d = conv_transpose2d_k2s2(latent, sd["decoder.0.weight"], sd["decoder.0.bias"], dtype=dtype)  # :63
    d = relu(_bn(sd, "decoder.1", d, dtype)); inter["dec1"] = d
    d = conv_transpose2d_k2s2(d, sd["decoder.3.weight"], sd["decoder.3.bias"], (0, 1), dtype)      # :68-69
    d = relu(_bn(sd, "decoder.4", d, dtype)); inter["dec2"] = d
    d = conv_transpose2d_k2s2(d, sd["decoder.6.weight"], sd["decoder.6.bias"], dtype=dtype)       # :74
    d = relu(_bn(sd, "decoder.7", d, dtype)); inter["dec3"] = d
    d = conv_transpose2d_k2s2(d, sd["decoder.9.weight"], sd["decoder.9.bias"], dtype=dtype)       # :79
    T = x.shape[1]
    Tr = d.shape[2]
    if Tr < T:                                                                  # :113-119
        d = np.concatenate([d, np.zeros(d.shape[:2] + (T - Tr, d.shape[3]), dtype)], axis=2)
    elif Tr > T:                                                                # :120-121
        d = d[:, :, :T, :]
    recon = d[:, 0]                                                             # :123
    if return_intermediates:
        return recon, latent, inter
    return recon, latent


def per_sample_mse(recon, x, dtype=np.float32):
    """MSELoss(reduction='none').view(B,-1).mean(1) -- src/evaluation_cae.py:52-53,
    src/hybrid_ensemble.py:55."""
    d = (np.asarray(recon, dtype) - np.asarray(x, dtype))
    return (d * d).reshape(d.shape[0], -1).mean(axis=1, dtype=dtype)


# --------------------------------------------------------------------------- losses / optimiser
def smooth_labels(y, eps):
    """y*(1-eps) + 0.5*eps -- src/train.py:311-315."""
    return np.asarray(y, np.float32) * np.float32(1.0 - eps) + np.float32(0.5 * eps)


def bce_with_logits(z, y, dtype=np.float64):
    """BCEWithLogitsLoss(mean): mean(max(z,0) - z*y + log1p(exp(-|z|))) -- src/train.py:317.
    Returns (loss, dloss/dz) with dz = (sigmoid(z) - y)/B."""
    z = np.asarray(z, dtype)
    y = np.asarray(y, dtype)
    loss = np.mean(np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z))))
    dz = (1.0 / (1.0 + np.exp(-z)) - y) / z.size
    return loss, dz


def adamw_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, wd=0.01):
    """One torch.optim.AdamW update (decoupled weight decay, bias-corrected) -- the optimiser
    built at src/train.py:321-330 with torch defaults.  step is 1-based.  Returns (p, m, v)."""
    p = np.asarray(p, np.float32).copy()
    g = np.asarray(g, np.float32)
    p *= np.float32(1.0 - lr * wd)
    m = (np.float32(b1) * m + np.float32(1 - b1) * g).astype(np.float32)
    v = (np.float32(b2) * v + np.float32(1 - b2) * g * g).astype(np.float32)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = (np.sqrt(v) / np.float32(np.sqrt(bc2)) + np.float32(eps)).astype(np.float32)
    p -= (np.float32(lr / bc1) * m / denom).astype(np.float32)
    return p, m, v


# --------------------------------------------------------------------------- scorer / host-side pieces
def calculate_eer(scores, labels):
    """Equal error rate -- scripts/evaluation.py:7-39 (== src/evaluation.py:12-48).

    argsort ascending; FAR = [1, (n_spoof - cumsum(label==0))/n_spoof]; FRR = [0, cumsum(label==1)/n_bona];
    idx = argmin |FAR-FRR|; eer = mean of the two at idx; threshold rule :31-37; (0,0) if a class is empty.
    """
    s = np.array(scores)
    l = np.array(labels)
    order = np.argsort(s)
    ss, sl = s[order], l[order]
    n_bona = np.sum(l)
    n_spoof = len(l) - n_bona
    if n_bona == 0 or n_spoof == 0:
        return 0.0, 0.0
    far = np.concatenate([[1.0], (n_spoof - np.cumsum(sl == 0)) / n_spoof])
    frr = np.concatenate([[0.0], np.cumsum(sl == 1) / n_bona])
    i = int(np.argmin(np.abs(far - frr)))
    eer = (far[i] + frr[i]) / 2.0
    if i == 0:
        thr = ss[0] - 1e-6
    elif i == len(ss):
        thr = ss[-1] + 1e-6
    else:
        thr = ss[i - 1]
    return float(eer), float(thr)


def confusion_at_threshold(scores, labels, threshold):
    """scripts/evaluation.py:42-56 -> (tp, fp, tn, fn, far, frr); pred = score > threshold."""
    s = np.array(scores)
    l = np.array(labels).astype(int)
    pred = (s > threshold).astype(int)
    tp = int(np.sum((pred == 1) & (l == 1)))
    fn = int(np.sum((pred == 0) & (l == 1)))
    fp = int(np.sum((pred == 1) & (l == 0)))
    tn = int(np.sum((pred == 0) & (l == 0)))
    far = fp / (fp + tn) if (fp + tn) > 0 else 0.0
    frr = fn / (tp + fn) if (tp + fn) > 0 else 0.0
    return tp, fp, tn, fn, float(far), float(frr)


def normalizer_fit(feature_list):
    """FeatureNormalizer.fit -- src/dataset_cae.py:29-35: per-feature mean and *unbiased* std over
    all frames of all (T,F) tensors, std clamped to >= 1e-8."""
    a = np.concatenate([np.asarray(f, np.float32) for f in feature_list], axis=0)
    mean = a.mean(axis=0, dtype=np.float64).astype(np.float32)
    std = np.maximum(a.std(axis=0, ddof=1, dtype=np.float64).astype(np.float32), np.float32(1e-8))
    return mean, std


def normalizer_transform(x, mean, std):
    """FeatureNormalizer.transform -- src/dataset_cae.py:37-41."""
    return ((np.asarray(x, np.float32) - mean) / std).astype(np.float32)


def normalise_scores(s):
    """Min-max to [0,1]; zeros if range < 1e-12 -- src/hybrid_ensemble.py:64-69."""
    s = np.asarray(s)
    lo, hi = s.min(), s.max()
    if hi - lo < 1e-12:
        return np.zeros_like(s)
    return (s - lo) / (hi - lo)


def hybrid_alpha_sweep(sup_scores, cae_scores, labels, alpha_steps=21):
    """src/hybrid_ensemble.py:139-151: alpha in linspace(0,1,n); alpha*sup_norm+(1-alpha)*cae_norm;
    strict '<' keeps the first best.  Returns (table [(alpha, eer)], best_eer, best_alpha)."""
    sn, cn = normalise_scores(np.asarray(sup_scores)), normalise_scores(np.asarray(cae_scores))
    table, best_eer, best_alpha = [], 1.0, 0.0
    for a in np.linspace(0.0, 1.0, alpha_steps):
        eer, _ = calculate_eer((a * sn + (1 - a) * cn).tolist(), list(labels))
        table.append((float(a), eer))
        if eer < best_eer:
            best_eer, best_alpha = eer, float(a)
    return table, best_eer, best_alpha


def ensemble_mean(score_lists):
    """np.mean over models of sigmoid scores -- src/ensemble.py:121."""
    return np.mean([np.asarray(s) for s in score_lists], axis=0)
