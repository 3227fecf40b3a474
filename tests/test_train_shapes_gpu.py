"""Round 3: training-step parity at the REAL frame count and at odd / ragged shapes.

Rounds 1-2 pinned the CNN2D gradients to the reference only at [4,16,180] and [16,64,180] (powers of two below the pools); here
the same quantities -- logits, loss, every parameter gradient, the BatchNorm running statistics and the parameters after one
AdamW step -- are held
  * to the reference's OWN autograd / AdamW results at [2,321,180] (tests/golden/*_train_t321.npz, made by
    tests/golden/make_golden_r3.py running src/model.py / src/model_cnn1d.py): the (2,1) floor pools drop frame 320
    (src/model.py:18,24), H1 = 160, H2 = 80, six 30-column strips on F = 180;
  * to the float64 training oracle (oracle/torch_ref.py, pinned to those goldens on the CPU by tests/test_oracle_golden.py) at
    T = 321, 322, 323 (T mod 4 = 1, 2, 3) with F = 180 and at the ragged shapes [3,21,65], [2,33,5]: fp32 mode against the
    un-rounded oracle, bf16 mode (matrix-core AND vector block-1 passes, with and without a folded augmentation) against the
    rounding-faithful one at 2 % of each gradient's scale.

TOLERANCE, fp32 mode.  The [4,16,180] bound is 2e-4 * scale per element.  At [B,321,180] a block holds B * 1.8 M ReLU inputs; an
implementation that rounds differently from the comparison target (fp32 accumulation order vs float64; the reference's own fp32
vs exact arithmetic) moves pre-activations by ~1e-7 .. 1e-6, so ON AVERAGE ONE ELEMENT PER BLOCK takes the other side of its ReLU.
At batch 2 one such element carries |dy| ~ 1e-3: the reference's OWN float32 gradients sit 3.3e-3 (conv.6.bias), 2.7e-3
(conv.5.weight), 1.1e-3 (conv.1.weight), 4e-4 (conv.0.weight) of their scales from float64 arithmetic for exactly that reason
(measured in the build container: one flipped block-2 element caused by native-BatchNorm rounding in block 1; with torch's own
float32 batch_norm in place of the formula the restatement matches the golden to 0.0; block 3 and the classifier agree to 6e-6).
One block-3 element at batch 2 is worth 1.3e-2 of conv.10.weight's scale (measured on the GPU at [2,323,180]; 1.6e-4 at [3,322,180]
where no element flipped, 1e-6 at the small shapes).  The perturbation is sparse and bounded, so the tests hold: every element within 3e-2 * scale, every gradient tensor within 3e-3
in relative L2 norm, and logits / loss / running statistics at the fixture's tight bounds (they do not depend on a mask's sign).
"""
import math
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NOISE2D = ("conv.0.bias", "conv.5.bias", "conv.10.bias")     # bias in front of a batch-statistics BatchNorm: zero gradient
NOISE1D = ("conv.0.bias", "conv.4.bias", "conv.8.bias")
LOOSE, L2TOL = 3e-2, 3e-3


def _to_np(v):
    return v.detach().float().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)


def _close_up_to_relu_flips(got, want, name, loose=LOOSE, l2tol=L2TOL, log=None):
    got, want = _to_np(got).astype(np.float64), _to_np(want).astype(np.float64)
    scale = max(np.abs(want).max(), 1e-6)
    d = np.abs(got - want)
    rel_l2 = float(np.sqrt((d * d).sum() / max((want * want).sum(), 1e-30)))
    if log is not None:
        log.append((name, float(d.max() / scale), rel_l2, d.size))
    assert d.max() <= loose * scale, (name, float(d.max() / scale))
    assert rel_l2 <= l2tol, (name, rel_l2)


def _check_grads(named_grads, want, noise, log=None, **kw):
    for name, got in named_grads:
        if name in noise:
            floor = 1e-4 * float(np.abs(_to_np(want[name.replace("bias", "weight")])).max()) + 1e-6
            assert float(got.abs().max()) < floor, (name, float(got.abs().max()), floor)
            continue
        _close_up_to_relu_flips(got, want[name], name, log=log, **kw)


def _check_state(sd, want, init, noise, steps=1, log=None):
    for k, v in sd.items():
        w = want[k]
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(w), k
        elif k in noise:              # Adam normalises the rounding noise of a zero gradient to +-lr per step: only the bound holds
            assert np.abs(_to_np(v) - _to_np(init[k])).max() <= steps * 1e-3 * 1.02 + 1e-6, k
        elif k.endswith("running_mean"):   # inherits the bias noise above
            np.testing.assert_allclose(_to_np(v), _to_np(w), atol=1e-5 + steps * 1e-3, rtol=2e-4, err_msg=k)
        elif k.endswith("running_var"):
            np.testing.assert_allclose(_to_np(v), _to_np(w), atol=1e-6, rtol=2e-4, err_msg=k)
        else:
            # AdamW's first step moves every element by lr * g / (|g| + 1e-8), i.e. by +-lr whatever |g| is: an element whose
            # gradient is rounding noise around zero may take the other sign (2 * lr apart).  All elements within 2 * lr; all but
            # max(3, 3 %) of them within the [4,16,180] fixture's bound.
            d = np.abs(_to_np(v).astype(np.float64) - _to_np(w).astype(np.float64))
            nbad = int((d > 2e-5 + 2e-4 * np.abs(_to_np(w))).sum())
            if log is not None:
                log.append(("after1." + k, float(d.max()), nbad / d.size, d.size))
            assert d.max() <= steps * 2.05e-3, (k, float(d.max()))
            assert nbad <= max(3, math.ceil(0.03 * d.size)), (k, nbad, d.size)


def _print_log(tag, log):
    worst = sorted(log, key=lambda r: -r[1])[:4]
    print(f"[{tag}] worst: " + "; ".join(f"{n} max {e:.2e} (l2/frac {b:.2e}, n={s})" for n, e, b, s in worst))


def _cnn2d(sd, F, precision="fp32"):
    from dfa_amd.model import CNN2D
    m = CNN2D(in_features=F, dropout=0.0, precision=precision)
    m.load_state_dict({k: (v.clone() if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in sd.items()})
    return m.to("cuda").train()


def _random_cnn2d_state(F, seed):
    """A CNN2D state with non-trivial BatchNorm affine parameters and a classifier that gives logits of order 1."""
    from dfa_amd.model import CNN2D
    torch.manual_seed(seed)
    m = CNN2D(in_features=F, dropout=0.0)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for i in (1, 6, 11):
            m.conv[i].weight.copy_(0.5 + torch.rand(m.conv[i].weight.shape, generator=g))
            m.conv[i].bias.copy_(0.1 * torch.randn(m.conv[i].bias.shape, generator=g))
        m.classifier.weight.mul_(40.0 * math.sqrt(180.0 / F))
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


# --------------------------------------------------------------------------------------------- vs the reference's own results
def test_cnn2d_train_step_at_321_frames_matches_reference_autograd(golden):
    from dfa_amd.training.train_step import NativeTrainer
    _, g = golden("cnn2d_train_t321")
    init = {k[len("init.sd."):]: v for k, v in g.items() if k.startswith("init.sd.")}
    want_g = {k[len("grad."):]: v for k, v in g.items() if k.startswith("grad.")}
    want_s = {k[len("after1."):]: v for k, v in g.items() if k.startswith("after1.")}
    eps = float(g["label_smoothing"])
    x = torch.from_numpy(g["x"]).to("cuda").transpose(1, 2)          # the strided view src/train.py:65 feeds
    y = torch.from_numpy(g["y"]).to("cuda")
    log = []
    # (a) drop-in path: torch criterion + torch.optim.AdamW over the autograd bridge (src/train.py:71-76)
    m = _cnn2d(init, 180)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
    logits = m(x).squeeze(-1)
    loss = torch.nn.BCEWithLogitsLoss()(logits, y * (1 - eps) + 0.5 * eps)
    opt.zero_grad()
    loss.backward()
    np.testing.assert_allclose(_to_np(logits), g["logits"], atol=2e-4, rtol=1e-5)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    _check_grads([(n, p.grad) for n, p in m.named_parameters()], want_g, NOISE2D, log=log)
    opt.step()
    _check_state(m.state_dict(), want_s, init, NOISE2D, log=log)
    # (b) the all-C-ABI trainer (fused BCE, flat gradient, fused AdamW)
    m2 = _cnn2d(init, 180)
    tr = NativeTrainer(m2, lr=1e-3, weight_decay=0.01, label_smoothing=eps)
    loss2 = tr.step(x, y)
    np.testing.assert_allclose(loss2.item(), g["loss"], rtol=1e-5)
    _check_state(m2.state_dict(), want_s, init, NOISE2D, log=log)
    _print_log("cnn2d t321 vs reference", log)


def test_cnn1d_train_step_at_321_frames_matches_reference_autograd(golden):
    from dfa_amd.model_cnn1d import CNN1D
    _, g = golden("cnn1d_train_t321")
    init = {k[len("init.sd."):]: v for k, v in g.items() if k.startswith("init.sd.")}
    want_g = {k[len("grad."):]: v for k, v in g.items() if k.startswith("grad.")}
    want_s = {k[len("after1."):]: v for k, v in g.items() if k.startswith("after1.")}
    eps = float(g["label_smoothing"])
    m = CNN1D(in_features=180, dropout=0.0)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in init.items()})
    m = m.to("cuda").train()
    x = torch.from_numpy(g["x"]).to("cuda").transpose(1, 2)
    y = torch.from_numpy(g["y"]).to("cuda")
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
    logits = m(x).squeeze(-1)
    loss = torch.nn.BCEWithLogitsLoss()(logits, y * (1 - eps) + 0.5 * eps)
    opt.zero_grad()
    loss.backward()
    log = []
    np.testing.assert_allclose(_to_np(logits), g["logits"], atol=2e-4, rtol=1e-5)
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    _check_grads([(n, p.grad) for n, p in m.named_parameters()], want_g, NOISE1D, log=log)
    opt.step()
    _check_state(m.state_dict(), want_s, init, NOISE1D, log=log)
    _print_log("cnn1d t321 vs reference", log)


# --------------------------------------------------------------------------------------------- vs the float64 oracle, fp32 mode
SHAPES = [(4, 321, 180), (3, 322, 180), (2, 323, 180), (3, 21, 65), (2, 33, 5)]


@pytest.mark.parametrize("B,T,F", SHAPES)
def test_cnn2d_fp32_train_step_matches_oracle_at_odd_shapes(B, T, F):
    from oracle import torch_ref as R
    sd = _random_cnn2d_state(F, seed=100 + T + F)
    gen = torch.Generator().manual_seed(T * 1000 + F)
    stored = (torch.randn(B, F, T, generator=gen) * 3.2 - 0.07)
    y = (torch.rand(B, generator=gen) > 0.5).float()
    y[0] = 1.0 - y[1] if B > 1 else y[0]
    eps = 0.05
    logits_w, loss_w, grads_w, stats = R.cnn2d_train_step_emulated(sd, stored.transpose(1, 2), y, eps, emulate=None, return_stats=True)
    after_w = R.state_after_adamw_step(sd, grads_w, stats)
    m = _cnn2d(sd, F)
    x = stored.to("cuda").transpose(1, 2)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
    logits = m(x).squeeze(-1)
    loss = torch.nn.BCEWithLogitsLoss()(logits, y.to("cuda") * (1 - eps) + 0.5 * eps)
    opt.zero_grad()
    loss.backward()
    log = []
    np.testing.assert_allclose(_to_np(logits), logits_w.numpy(), atol=2e-4, rtol=1e-5)
    np.testing.assert_allclose(loss.item(), loss_w, rtol=1e-5)
    _check_grads([(n, p.grad) for n, p in m.named_parameters()], grads_w, NOISE2D, log=log)
    opt.step()
    _check_state(m.state_dict(), after_w, sd, NOISE2D, log=log)
    _print_log(f"cnn2d fp32 [{B},{T},{F}] vs float64 oracle", log)


@pytest.mark.parametrize("B,T,F", [(3, 321, 180), (2, 37, 180)])
def test_cnn1d_fp32_train_step_matches_oracle(B, T, F):
    from dfa_amd.model_cnn1d import CNN1D
    from oracle import torch_ref as R
    torch.manual_seed(5 + T)
    m = CNN1D(in_features=F, dropout=0.0)
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for i in (1, 5, 9):
            m.conv[i].weight.copy_(0.5 + torch.rand(m.conv[i].weight.shape, generator=g))
            m.conv[i].bias.copy_(0.1 * torch.randn(m.conv[i].bias.shape, generator=g))
        m.classifier.weight.mul_(40.0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    stored = torch.randn(B, F, T, generator=g) * 3.2 - 0.07
    y = (torch.rand(B, generator=g) > 0.5).float()
    logits_w, loss_w, grads_w, stats = R.cnn1d_train_step(sd, stored.transpose(1, 2), y, 0.05, return_stats=True)
    after_w = R.state_after_adamw_step(sd, grads_w, stats)
    m = m.to("cuda").train()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
    logits = m(stored.to("cuda").transpose(1, 2)).squeeze(-1)
    loss = torch.nn.BCEWithLogitsLoss()(logits, y.to("cuda") * 0.95 + 0.025)
    opt.zero_grad()
    loss.backward()
    log = []
    np.testing.assert_allclose(_to_np(logits), logits_w.numpy(), atol=2e-4, rtol=1e-5)
    np.testing.assert_allclose(loss.item(), loss_w, rtol=1e-5)
    _check_grads([(n, p.grad) for n, p in m.named_parameters()], grads_w, NOISE1D, log=log)
    opt.step()
    _check_state(m.state_dict(), after_w, sd, NOISE1D, log=log)
    _print_log(f"cnn1d fp32 [{B},{T},{F}] vs float64 oracle", log)


# --------------------------------------------------------------------------------------------- bf16 mode vs the rounding-faithful oracle
@pytest.mark.parametrize("B,T,F", SHAPES)
@pytest.mark.parametrize("path", ["conv1_mfma", "conv1_vector", "augment_folded", "augment_folded_jitter"])
def test_cnn2d_bf16_train_step_matches_emulated_oracle_at_odd_shapes(B, T, F, path):
    """bf16 storage mode: every gradient within 2 % of its scale of oracle.torch_ref.cnn2d_train_step_emulated (bf16 rounding at
    the kernels' storage points), for the matrix-core block-1 passes, the vector ones, and a batch whose augmentation (time /
    feature masks, roll, channel drop -- and, last case, jitter noise) is folded into the loads of the kernels that read x."""
    from dfa_amd import _lib
    from dfa_amd.augmentation import FusedAugment
    from oracle import torch_ref as R
    sd = _random_cnn2d_state(F, seed=300 + T + F)
    gen = torch.Generator().manual_seed(T * 1000 + F + 7)
    stored = (torch.randn(B, F, T, generator=gen) * 3.2 - 0.07).to(torch.bfloat16)
    y = (torch.rand(B, generator=gen) > 0.5).float()
    x = stored.to("cuda").transpose(1, 2)
    ctx = _lib.Context.get(x.device)
    m = _cnn2d(sd, F, precision="bf16")
    x_oracle, round_x = stored.float().transpose(1, 2), True
    try:
        ctx.set_option("conv1_mfma", 0 if path == "conv1_vector" else 1)
        if path.startswith("augment"):
            jitter = path.endswith("jitter")
            cfg = dict(spec_augment=True, time_mask_ratio=0.2, feature_mask=True, feature_mask_ratio=0.1, time_shift=True,
                       time_shift_ratio=0.1, channel_drop=True, channel_drop_prob=0.3, gaussian_jitter=jitter,
                       gaussian_jitter_std=0.05)
            random.seed(41); torch.manual_seed(41)
            x_aug = FusedAugment(seed=5, fold=False, out_dtype=torch.float32, **cfg)(x)     # the stand-alone pass (pinned to the reference's
            x_oracle, round_x = x_aug.float().cpu(), not jitter                              # augmentation fixtures elsewhere) feeds the oracle
            random.seed(41); torch.manual_seed(41)
            FusedAugment(seed=5, fold=True, **cfg)(x)                                        # arms the same draw for the next forward
        logits = m(x).squeeze(-1)
        loss = torch.nn.BCEWithLogitsLoss()(logits, y.to("cuda") * 0.95 + 0.025)
        loss.backward()
    finally:
        ctx.set_option("conv1_mfma", 1)
    logits_w, loss_w, emu = R.cnn2d_train_step_emulated(sd, x_oracle, y, 0.05, "bf16", round_x=round_x)
    _, _, ref = R.cnn2d_train_step_emulated(sd, x_oracle, y, 0.05, None)
    assert abs(loss.item() - loss_w) < 2e-2 * max(1.0, abs(loss_w)), (loss.item(), loss_w)
    worst = ("", 0.0)
    for name, p in m.named_parameters():
        if name in NOISE2D:
            continue
        scale = max(float(ref[name].abs().max()), 1e-6)
        rel = float((p.grad.float().cpu() - emu[name]).abs().max()) / scale
        worst = max(worst, (name, rel), key=lambda r: r[1])
        # 2 % at the 321-frame shapes (measured 0.4 .. 1.0 %).  At the two tiny shapes a block holds ~1000 positions per channel, so
        # the handful of stored activations that round to the neighbouring bf16 value (fp32 accumulation here, float64 in the
        # oracle) weigh ten times more: measured up to 2.9 % (conv.10.weight, [3,21,65] with jitter) -> 4 %.
        assert rel < (0.02 if T * F > 4000 else 0.04), (name, (B, T, F), path, rel)
    print(f"[cnn2d bf16 {path} [{B},{T},{F}]] worst gradient {worst[0]} {worst[1]:.2e} of its scale")


# --------------------------------------------------------------------------------------------- ADVICE r2
def test_forced_time_split_at_full_batch_stays_inside_the_workspace():
    """a forced time_split on a large batch used to write the chunk slabs past the planned workspace.  The
    plan now reserves them whenever a split is forced; a canary behind the region dfa_workspace_bytes asks for must survive,
    and the logits must equal the unsplit ones bit for bit."""
    import ctypes as C
    from dfa_amd import _lib
    from dfa_amd.model import CNN2D
    torch.manual_seed(0)
    m = CNN2D(in_features=180, precision="bf16").to("cuda").eval()
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(256, 180, 321, generator=g) * 3.2).to("cuda", torch.bfloat16).transpose(1, 2)
    ctx = _lib.Context.get(x.device)
    base = m(x).clone()
    try:
        for forced in (2, 4):
            ctx.set_option("time_split", forced)
            prec = _lib.PRECISIONS["bf16"]
            nbytes = ctx.lib.dfa_workspace_bytes(ctx.handle, _lib.MODEL_CNN2D, 256, 321, 180, prec)
            ctx.set_option("time_split", -1)
            assert nbytes > ctx.lib.dfa_workspace_bytes(ctx.handle, _lib.MODEL_CNN2D, 256, 321, 180, prec)
            ctx.set_option("time_split", forced)
            pad = 1 << 20
            buf = torch.full((nbytes + pad,), 0x5A, dtype=torch.uint8, device="cuda")
            logits = torch.empty((256, 1), device="cuda")
            sb, st, sf = x.stride()
            m._ensure_prepared(ctx)
            _lib.check(ctx.handle, ctx.lib.dfa_cnn2d_forward(
                ctx.handle, C.c_void_p(x.data_ptr()), _lib.DTYPE_BF16, 256, 321, 180, sb, st, sf, C.c_void_p(logits.data_ptr()),
                None, C.c_void_p(buf.data_ptr()), nbytes))
            torch.cuda.synchronize()
            assert bool((buf[nbytes:] == 0x5A).all()), f"forced split {forced}: bytes behind the workspace were overwritten"
            assert torch.equal(logits, base), forced
            # a workspace sized for the automatic plan is refused instead of overrun
            ctx.set_option("time_split", -1)
            small = ctx.lib.dfa_workspace_bytes(ctx.handle, _lib.MODEL_CNN2D, 256, 321, 180, prec)
            ctx.set_option("time_split", forced)
            code = ctx.lib.dfa_cnn2d_forward(
                ctx.handle, C.c_void_p(x.data_ptr()), _lib.DTYPE_BF16, 256, 321, 180, sb, st, sf, C.c_void_p(logits.data_ptr()),
                None, C.c_void_p(buf.data_ptr()), small)
            assert code == _lib.E_WORKSPACE
    finally:
        ctx.set_option("time_split", -1)


def test_misaligned_embedding_pointer_is_refused():
    import ctypes as C
    from dfa_amd import _lib
    from dfa_amd.model import CNN2D
    m = CNN2D(in_features=180, precision="bf16").to("cuda").eval()
    x = torch.randn(2, 321, 180, device="cuda")
    ctx = _lib.Context.get(x.device)
    m(x)
    nbytes = ctx.lib.dfa_workspace_bytes(ctx.handle, _lib.MODEL_CNN2D, 2, 321, 180, _lib.PRECISIONS["bf16"])
    ws = ctx.workspace(nbytes)
    emb = torch.empty(2 * 128 * 180 + 4, device="cuda")
    logits = torch.empty((2, 1), device="cuda")
    code = ctx.lib.dfa_cnn2d_forward(ctx.handle, C.c_void_p(x.data_ptr()), _lib.DTYPE_F32, 2, 321, 180, *x.stride(),
                                     C.c_void_p(logits.data_ptr()), C.c_void_p(emb.data_ptr() + 4), C.c_void_p(ws.data_ptr()), ws.numel())
    assert code == _lib.E_BAD_SHAPE and b"16-byte" in ctx.lib.dfa_last_error(ctx.handle)


def test_failed_forward_disarms_the_folded_augmentation():
    """an armed one-shot augmentation must not survive a forward_train that fails its argument checks."""
    from dfa_amd.augmentation import FusedAugment
    sd = _random_cnn2d_state(180, seed=1)
    m = _cnn2d(sd, 180)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 40, 180, generator=g).to("cuda")
    plain = m(x).detach().clone()
    random.seed(3); torch.manual_seed(3)
    FusedAugment(seed=1, fold=True, spec_augment=True, time_mask_ratio=0.3, time_shift=True, time_shift_ratio=0.2)(x)
    with pytest.raises(ValueError):
        m(torch.randn(2, 48, 180, device="cuda"))          # other T than the armed draw: refused, and the arm is gone
    again = m(x).detach()
    assert torch.equal(again, plain)


def test_flat_adamw_resume_keeps_the_reduced_lr():
    """load_state_dict must also restore the lr of the stand-in optimiser the plateau scheduler drives."""
    from dfa_amd.training.train_step import NativeTrainer
    sd = _random_cnn2d_state(180, seed=4)
    tr = NativeTrainer(_cnn2d(sd, 180), lr=1e-3)
    sched = tr.plateau_scheduler(mode="min", factor=0.5, patience=0)
    sched.step(1.0); sched.step(2.0)                        # no improvement -> lr halves
    assert abs(tr.lr - 5e-4) < 1e-12
    saved = tr.state_dict()
    tr2 = NativeTrainer(_cnn2d(sd, 180), lr=1e-3)
    sched2 = tr2.plateau_scheduler(mode="min", factor=0.5, patience=3)
    tr2.load_state_dict(saved)
    sched2.step(1.0)                                        # an improving epoch must not undo the restored lr
    assert abs(tr2.lr - 5e-4) < 1e-12
