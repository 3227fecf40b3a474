"""Round 3: the all-C-ABI (no-autograd) trainers for the auto-encoder and CNN1D, dfa_mse_fwd_bwd, and the flat-gradient sink of
the autograd bridges (VERDICT r2 missing #2 / weak #11; SURVEY 8(b) "Train" exports)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CAE_NOISE = {f"encoder.{i}.bias" for i in (0, 4, 8, 12)} | {f"decoder.{i}.bias" for i in (0, 3, 6)}


def _cae(g, precision="fp32"):
    from dfa_amd.model_cae import ConvAutoencoder
    m = ConvAutoencoder(precision=precision)
    m.load_state_dict({k[len("init.sd."):]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("init.sd.")})
    return m.to("cuda").train()


def test_mse_fwd_bwd_matches_torch():
    """dfa_mse_fwd_bwd = nn.MSELoss()(recon, x) and its autograd gradient (src/train_cae.py:67-68,203), any x layout / dtype."""
    from dfa_amd import _lib
    g = torch.Generator().manual_seed(0)
    ctx = _lib.Context.get(torch.device("cuda", 0))
    for (B, T, F), xdt, view in (((3, 321, 180), torch.float32, False), ((2, 37, 20), torch.bfloat16, True), ((1, 5, 4), torch.float32, True)):
        recon = torch.randn(B, T, F, generator=g).to("cuda")
        stored = torch.randn(B, F, T, generator=g) if view else torch.randn(B, T, F, generator=g)
        x = stored.to("cuda", xdt)
        x = x.transpose(1, 2) if view else x
        loss = torch.zeros(1, device="cuda")
        d = torch.empty_like(recon)
        ctx.use_current_stream()
        _lib.check(ctx.handle, ctx.lib.dfa_mse_fwd_bwd(ctx.handle, C.c_void_p(recon.data_ptr()), C.c_void_p(x.data_ptr()),
                                                      _lib.x_dtype_code(x), B, T, F, *x.stride(), C.c_void_p(loss.data_ptr()),
                                                      C.c_void_p(d.data_ptr())))
        r = recon.clone().requires_grad_(True)
        want = torch.nn.MSELoss()(r, x.float())
        want.backward()
        np.testing.assert_allclose(loss.item(), want.item(), rtol=2e-6)
        assert torch.allclose(d, r.grad, rtol=1e-6, atol=1e-9)
        # loss only / gradient only
        _lib.check(ctx.handle, ctx.lib.dfa_mse_fwd_bwd(ctx.handle, C.c_void_p(recon.data_ptr()), C.c_void_p(x.data_ptr()),
                                                      _lib.x_dtype_code(x), B, T, F, *x.stride(), C.c_void_p(loss.data_ptr()), None))
        np.testing.assert_allclose(loss.item(), want.item(), rtol=2e-6)
    assert ctx.lib.dfa_mse_fwd_bwd(ctx.handle, None, None, 0, 1, 1, 1, 1, 1, 1, None, None) == _lib.E_NULL_PTR


def test_cae_native_trainer_matches_reference_and_autograd_bridge(golden):
    """CaeNativeTrainer.step (forward writes only mse[B]; dfa_cae_backward(drecon = NULL) forms 2 (recon - x) / N in its first
    kernel; gradients land in the flat views) against the reference's own loss / gradients / post-AdamW parameters, and against
    the autograd bridge + nn.MSELoss on the same weights."""
    from dfa_amd.training.train_step import CaeNativeTrainer
    _, g = golden("cae_train")
    x = torch.from_numpy(g["ls0.x"]).to("cuda")
    m = _cae(g)
    tr = CaeNativeTrainer(m, lr=1e-3, weight_decay=0.01)
    loss = tr.step(x)
    np.testing.assert_allclose(loss.item(), g["ls0.loss"], rtol=2e-5)
    ref = _cae(g)
    recon, _ = ref(x)
    torch.nn.MSELoss()(recon, x).backward()
    for (name, p), gv in zip(ref.named_parameters(), tr.grad_views):
        want = g[f"ls0.grad.{name}"]
        if name in CAE_NOISE:
            continue
        scale = max(np.abs(want).max(), 1e-6)
        np.testing.assert_allclose(gv.cpu().numpy(), want, atol=3e-4 * scale + 1e-7, rtol=5e-3, err_msg=name)
        # same kernels behind both paths; only the upstream gradient is recomputed (fma order): far below the reference bound
        assert float((gv - p.grad).abs().max()) <= 2e-5 * scale + 1e-8, name
    for k, v in m.state_dict().items():
        want = g[f"ls0.after1.{k}"]
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(want), k
        elif k in CAE_NOISE:
            assert np.abs(v.cpu().numpy() - g["init.sd." + k]).max() <= 1.02e-3 + 1e-6
        elif k.endswith("running_mean"):
            np.testing.assert_allclose(v.cpu().numpy(), want, atol=1.1e-3, rtol=2e-4, err_msg=k)
        else:
            got = v.cpu().numpy()
            close = np.isclose(got, want, atol=3e-5, rtol=3e-4)
            assert close.mean() > 0.9999, (k, 1.0 - close.mean())
            assert np.abs(got - want).max() <= 1.05e-3, k
    assert tr.flat_g.numel() == 561_633
    # bf16 storage mode, bf16 features, real frame count: loss equals the autograd bridge's, and training goes down
    m16, r16 = _cae(g, "bf16"), _cae(g, "bf16")
    gen = torch.Generator().manual_seed(3)
    xb = torch.randn(4, 321, 180, generator=gen).to("cuda", torch.bfloat16)
    t16 = CaeNativeTrainer(m16, lr=1e-3, weight_decay=0.0)
    l0 = t16.step(xb).item()
    rr, _ = r16(xb)
    np.testing.assert_allclose(l0, torch.nn.MSELoss()(rr, xb.float()).item(), rtol=1e-5)
    for _ in range(5):
        l1 = t16.step(xb).item()
    assert np.isfinite(l1) and l1 < l0


def test_cnn1d_native_trainer_matches_reference(golden):
    """NativeTrainer on a CNN1D (forward_train -> fused BCE -> backward into the flat views -> fused AdamW) against the reference's
    own AdamW results at [2,321,180] (tests/golden/cnn1d_train_t321.npz)."""
    from dfa_amd.model_cnn1d import CNN1D
    from dfa_amd.training.train_step import NativeTrainer
    _, g = golden("cnn1d_train_t321")
    m = CNN1D(in_features=180, dropout=0.0)
    m.load_state_dict({k[len("init.sd."):]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("init.sd.")})
    m = m.to("cuda").train()
    tr = NativeTrainer(m, lr=1e-3, weight_decay=0.01, label_smoothing=float(g["label_smoothing"]))
    assert tr.kind == "cnn1d" and tr.flat_g.numel() == 48_801
    x = torch.from_numpy(g["x"]).to("cuda").transpose(1, 2)
    loss = tr.step(x, torch.from_numpy(g["y"]))
    np.testing.assert_allclose(loss.item(), g["loss"], rtol=1e-5)
    for (name, _), gv in zip(m.named_parameters(), tr.grad_views):
        if name in ("conv.0.bias", "conv.4.bias", "conv.8.bias"):
            continue
        want = g["grad." + name]
        np.testing.assert_allclose(gv.cpu().numpy(), want, atol=2e-4 * max(np.abs(want).max(), 1e-6) + 1e-7, rtol=2e-3, err_msg=name)
    for k, v in m.state_dict().items():
        want = g["after1." + k]
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(want), k
        elif k in ("conv.0.bias", "conv.4.bias", "conv.8.bias"):
            assert np.abs(v.cpu().numpy() - g["init.sd." + k]).max() <= 1.02e-3 + 1e-6
        elif k.endswith("running_mean"):
            np.testing.assert_allclose(v.cpu().numpy(), want, atol=1.1e-3, rtol=2e-4, err_msg=k)
        else:
            got = v.cpu().numpy()
            close = np.isclose(got, want, atol=2e-5, rtol=2e-4)
            assert close.mean() > 0.999, (k, 1.0 - close.mean())
            assert np.abs(got - want).max() <= 2.05e-3, k


def test_flat_trainer_bridge_writes_gradients_straight_into_the_flat_buffer(golden):
    """FlatTrainer + autograd bridge (arbitrary torch loss): the C-ABI backward writes into the flat views, autograd receives no
    per-parameter tensors to add; gradients equal the plain-autograd ones bit for bit; a detached .grad falls back cleanly."""
    from dfa_amd.training.train_step import FlatTrainer
    _, g = golden("cae_train")
    x = torch.from_numpy(g["ls0.x"]).to("cuda")
    plain = _cae(g)
    r, _ = plain(x)
    torch.nn.MSELoss()(r, x).backward()
    m = _cae(g)
    tr = FlatTrainer(m, lr=1e-3, weight_decay=0.01)
    tr.flat_g.fill_(123.0)                                   # stale content must be overwritten, not accumulated into
    tr.zero_grad()
    r2, _ = m(x)
    torch.nn.MSELoss()(r2, x).backward()
    for (n, p), q, gv in zip(m.named_parameters(), plain.parameters(), tr.grad_views):
        assert p.grad.data_ptr() == gv.data_ptr(), n
        assert torch.equal(p.grad, q.grad), n
    tr.step()
    # a parameter whose .grad was replaced: the bridge returns fresh tensors and autograd accumulates as usual
    first = next(m.parameters())
    first.grad = None
    r3, _ = m(x)
    torch.nn.MSELoss()(r3, x).backward()
    assert first.grad is not None and torch.isfinite(first.grad).all()


@pytest.mark.parametrize("option", ["cae_dgrad_mfma", "conv1_mfma", "dgrad_m16", "cae_conv_stats", "cae_bwd_fold", "cae_enc4_wide"])
@pytest.mark.parametrize("B,T,F", [(3, 96, 180), (2, 321, 180), (5, 48, 36)])
def test_cae_training_matrix_core_kernels_match_their_round2_twins(B, T, F, option):
    """Round 3, auto-encoder training step in bf16 mode (autograd of src/model_cae.py:40-79 inside loss.backward(),
    src/train_cae.py:71), each new kernel against the one it replaced (context option = 0):
    * cae_dgrad_mfma -- the three ConvTranspose2d data gradients on `convt_dgrad_bf16_kernel` (bf16 weights -- the ones the forward
      multiplied by -- fp32 accumulation, bf16 result in place) instead of an fp32-MFMA GEMM + cast pass.  Same forward, so the
      loss is identical; the gradients differ by the rounding of W in the backward (2^-9 per weight);
    * conv1_mfma -- block 1's statistics and backward passes on `conv1_mfma_kernel` (2 x 2-pool form of the CNN2D's matrix-core
      block 1, fused moment algebra) and its forward on `cae_enc1_mfma_kernel`, instead of three vector-ALU passes.  The forward
      differs by isolated bf16 ulps of the block-1 output, so the loss agrees to 2e-4.
    * dgrad_m16 -- the encoder's 64 -> 32 and 128 -> 64 data gradients on the 16x16x32 kernels of conv_split.hip (one launch each, as
      the CNN2D's) instead of the 32x32x16 forms (the second chained through fp32 partial sums): same products, other summation order.
    * cae_conv_stats -- encoder blocks 2-3 and decoder blocks 1-3 take their BatchNorm statistics in the convolution's epilogue (fp32 sums of the outputs before
      they are rounded for storage, as the CNN2D's blocks 2 / 3) instead of a separate pass over the stored bf16 output: mean and
      variance move by the (unbiased) storage rounding averaged over the batch, the loss agrees to 5e-4.
    * cae_enc4_wide -- encoder block 4 (128 -> 256) forward in one launch over all 128 input channels and its data gradient in two
      128-channel launches, instead of two / four 64-channel launches chained through fp32 partial sums: the same products in
      another summation order (the forward's output moves by isolated bf16 ulps, so it is held like the other forward-changing options).
    * cae_bwd_fold -- the decoder's BatchNorm-backward apply pass writes dz patch-major (no pixel-unshuffle pass) and sums the
      ConvTranspose2d bias gradient on the way (no channel-sum pass): the same dz values, so every gradient is bit-identical except
      those bias gradients (zero up to rounding; another summation order).
    Every gradient within 3 % relative L2 (the bound the emulated-oracle test gives decoder gradients; seven BatchNorm + ReLU
    layers amplify any re-rounding), convolution biases in front of a BatchNorm (gradient zero up to rounding) on their weight's
    scale, and the decoder's last block -- upstream of every changed kernel -- bit-identical for the data-gradient option."""
    from dfa_amd import _lib
    from dfa_amd.model_cae import ConvAutoencoder
    g = torch.Generator().manual_seed(B * 7 + T)
    x = torch.randn(B, T, F, generator=g).to("cuda", torch.bfloat16)
    ctx = _lib.Context.get(x.device)
    res, stats = {}, {}
    try:
        for arm in (0, 1):
            ctx.set_option(option, arm)
            torch.manual_seed(0)
            m = ConvAutoencoder(precision="bf16").to("cuda").train()
            recon, _ = m(x)
            loss = torch.nn.functional.mse_loss(recon, x.float())
            loss.backward()
            res[arm] = (float(loss), {n: p.grad.clone() for n, p in m.named_parameters()})
            stats[arm] = {n: b.clone() for n, b in m.named_buffers() if "running_" in n}
    finally:
        ctx.set_option(option, 1)
    if option == "cae_conv_stats":
        # the statistics themselves (through the running buffers one step updates): the epilogue's sums are those of the separate pass up to
        # the storage rounding of z (2^-9 per element, unbiased) -- a miscounted strip edge or row would show here at the 1e-3 .. 1e-2 level (0.1 / W of the variance)
        for n, b0 in stats[0].items():
            if n.startswith("encoder.1."):     # block 1: untouched
                assert torch.equal(stats[1][n], b0), n
            # (the decoder's layers also see the encoder's re-rounded output)
            assert float((stats[1][n] - b0).abs().max()) <= (1e-4 if n.startswith("encoder.") else 3e-4) * max(float(b0.abs().max()), 1.0), n
    forward_changes = option in ("conv1_mfma", "cae_conv_stats", "cae_enc4_wide")
    if option in ("cae_dgrad_mfma", "dgrad_m16", "cae_bwd_fold"):
        assert res[0][0] == res[1][0]
    else:
        # (statistics option: a layer with n pixels per channel moves its mean by ~2^-9 / sqrt(n) of a standard deviation -- n = 120 in
        # the last encoder block of the [5,48,36] batch: 2.2e-4 measured there, 7e-5 on [256,321,180])
        assert abs(res[0][0] - res[1][0]) <= (5e-4 if option == "cae_conv_stats" else 2e-4) * abs(res[0][0])
    worst = 0.0
    for n, g0 in res[0][1].items():
        g1 = res[1][1][n]
        assert torch.isfinite(g1).all(), n
        rel = float((g1 - g0).norm() / (g0.norm() + 1e-30))
        if option == "cae_dgrad_mfma" and n.startswith("decoder.9"):    # ConvTranspose2d(32 -> 1): untouched
            assert torch.equal(g0, g1), n
        if n.endswith(".bias") and n[:-5] + ".weight" in res[0][1] and res[0][1][n[:-5] + ".weight"].dim() == 4 and not n.startswith("decoder.9"):
            # a convolution bias in front of a BatchNorm: its gradient is zero up to rounding -- compare on the weight gradient's scale
            assert float((g1 - g0).abs().max()) <= 1e-2 * float(res[0][1][n[:-5] + ".weight"].abs().max()), n   # (bf16 storage noise of dz)
            continue
        if option == "cae_bwd_fold":
            assert torch.equal(g0, g1), n
        # end to end the block-1 and statistics options also change the forward (isolated bf16 ulps of a block's output): the encoder's gradients
        # then differ like any two valid bf16 implementations do -- the regime of the emulated-oracle test's 10 % bound (measured 7 % at [2,321,180], 10 % at the tiny [3,96,180]); the backward
        # pass itself is held to 1 % below, on one forward state
        assert rel <= (2e-1 if forward_changes else 3e-2), (n, rel)
        worst = max(worst, rel)
    print(f"cae {option} 1 vs 0 [{B},{T},{F}]: worst relative L2 over the gradients {worst:.2e}")
    if option != "conv1_mfma":
        return
    # ---- the block-1 backward pass in isolation: two backward calls on the SAME forward state, the option cleared in between
    # (dfa_cae_backward then runs the two vector-ALU passes on that state).  Block 1's gradients agree up to the pixels whose
    # pre-ReLU value is within rounding of zero, everything else is bit-identical.
    from dfa_amd.training.train_step import cae_forward_train_raw, cae_backward_raw
    torch.manual_seed(0)
    m = ConvAutoencoder(precision="bf16").to("cuda").train()
    names = [n for n, _ in m.named_parameters()]
    try:
        ctx.set_option("conv1_mfma", 1)
        _, _, _, c, ws, gen = cae_forward_train_raw(m, x, want_recon=False, want_latent=False, want_mse=True)
        ga = [torch.zeros_like(p) for p in m.parameters()]
        gb = [torch.zeros_like(p) for p in m.parameters()]
        cae_backward_raw(m, x, None, ga, c, ws, gen)
        ctx.set_option("conv1_mfma", 0)
        cae_backward_raw(m, x, None, gb, c, ws, gen)
    finally:
        ctx.set_option("conv1_mfma", 1)
    for n, a, b in zip(names, ga, gb):
        if n in ("encoder.0.weight", "encoder.1.weight", "encoder.1.bias"):
            scale = max(float(b.abs().max()), 1e-12)
            assert float((a - b).abs().max()) <= 1e-2 * scale, (n, float((a - b).abs().max()) / scale)
        elif n == "encoder.0.bias":
            assert float(a.abs().max()) <= 1e-3 * float(gb[0].abs().max()) + 1e-9, n
        else:
            assert torch.equal(a, b), n
