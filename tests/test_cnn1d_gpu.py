"""GPU parity tests for the CNN1D path (HIP kernels through the C ABI vs golden vectors / oracle)."""
import numpy as np
import pytest
import torch

from oracle import dfa_oracle as O

pytestmark = pytest.mark.gpu
TOL_F32 = 1e-4


def _model(sd):
    from dfa_amd.model_cnn1d import CNN1D
    m = CNN1D()
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.to("cuda").eval()


@pytest.mark.parametrize("tag", ["t321", "t64", "t7", "t16"])
def test_cnn1d_matches_golden(golden, tag):
    sd, g = golden("cnn1d_eval")
    model = _model(sd)
    stored = torch.from_numpy(g[f"{tag}.x_stored"]).to("cuda")
    logits = model(stored.transpose(1, 2))                       # strided view of the stored layout
    np.testing.assert_allclose(logits.cpu().numpy(), g[f"{tag}.logits"], atol=TOL_F32, rtol=0)
    logits_c = model(stored.transpose(1, 2).contiguous())       # other strides: the exact-fp32 fused kernel instead of the split-bf16 one
    np.testing.assert_allclose(logits_c.cpu().numpy(), g[f"{tag}.logits"], atol=TOL_F32, rtol=0)
    np.testing.assert_allclose(logits_c.cpu().numpy(), logits.cpu().numpy(), atol=5e-5, rtol=0)


def test_cnn1d_layers_match_golden(golden):
    sd, g = golden("cnn1d_eval")
    model = _model(sd)
    x = torch.from_numpy(g["t16.x_stored"]).to("cuda").transpose(1, 2)
    from dfa_amd import _lib
    ctx = _lib.Context.get(x.device)
    ctx.set_option("cnn1d_fused", 0)          # the three-launch path keeps h1 / h2 in the workspace (the fused kernel keeps them in LDS)
    try:
        model(x)
    finally:
        ctx.set_option("cnn1d_fused", 1)
    ws = ctx._ws
    T = 16
    n1 = 32 * T * 4
    off2 = (n1 + 255) // 256 * 256
    h1 = ws[:n1].view(torch.float32).view(1, 32, T).cpu().numpy()
    h2 = ws[off2:off2 + 64 * T * 4].view(torch.float32).view(1, 64, T).cpu().numpy()
    np.testing.assert_allclose(h1, g["t16.h1"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(h2, g["t16.h2"], atol=2e-5, rtol=1e-5)


def test_cnn1d_full_batch_vs_oracle_and_independence(golden):
    sd, _ = golden("cnn1d_eval")
    model = _model(sd)
    g = torch.Generator().manual_seed(3)
    stored = torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07
    full = model(stored.to("cuda").transpose(1, 2)).cpu().numpy()
    want = O.cnn1d_forward(sd, stored[:8].numpy().swapaxes(1, 2))
    np.testing.assert_allclose(full[:8], want, atol=TOL_F32, rtol=0)
    one = model(stored[200:201].to("cuda").transpose(1, 2)).cpu().numpy()
    np.testing.assert_allclose(one, full[200:201], atol=1e-6, rtol=0)


def test_cnn1d_errors(golden):
    sd, _ = golden("cnn1d_eval")
    model = _model(sd)
    with pytest.raises(ValueError):
        model(torch.zeros(2, 321, 100, device="cuda"))
    with pytest.raises(ValueError):
        model(torch.zeros(2, 321, 180, device="cuda", dtype=torch.bfloat16))
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 321, 180))


@pytest.mark.parametrize("B,T,F,layout", [(5, 321, 180, "bft_view"), (3, 321, 180, "btf"), (2, 1, 180, "bft_view"), (2, 2, 180, "btf"),
                                          (4, 33, 180, "bft_view"), (2, 384, 180, "bft_view"), (2, 385, 180, "bft_view"),
                                          (3, 97, 65, "btf"), (2, 40, 5, "bft_view"), (1, 64, 33, "bft_view")])
def test_cnn1d_fused_kernel_matches_three_launch_path_and_oracle(B, T, F, layout):
    """Round 3: the whole CNN1D forward as ONE kernel (csrc/cnn1d_fused.hip: fp32 matrix cores, activations in LDS, frame mean
    and classifier in the epilogue) against the three-launch path it replaces and the numpy oracle, over frame counts that are
    not multiples of the 32-frame tile, single frames, the largest T that fits LDS (384; 385 falls back), odd channel counts
    (zero-padded channel pair) and both input layouts.  fp32 fma chains on both sides: 1e-4 (the reference bar), measured ~1e-6."""
    from dfa_amd import _lib
    from dfa_amd.model_cnn1d import CNN1D
    torch.manual_seed(T + F)
    m = CNN1D(in_features=F)
    gen = torch.Generator().manual_seed(B * 1000 + T)
    with torch.no_grad():
        for i in (1, 5, 9):
            m.conv[i].running_mean.copy_(0.2 * torch.randn(m.conv[i].running_mean.shape, generator=gen))
            m.conv[i].running_var.copy_(0.5 + torch.rand(m.conv[i].running_var.shape, generator=gen))
            m.conv[i].weight.copy_(0.5 + torch.rand(m.conv[i].weight.shape, generator=gen))
            m.conv[i].bias.copy_(0.1 * torch.randn(m.conv[i].bias.shape, generator=gen))
        m.classifier.weight.mul_(20.0)
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    m = m.to("cuda").eval()
    stored = torch.randn(B, F, T, generator=gen) * 3.2 - 0.07 if layout == "bft_view" else torch.randn(B, T, F, generator=gen) * 3.2 - 0.07
    x = stored.to("cuda").transpose(1, 2) if layout == "bft_view" else stored.to("cuda")
    ctx = _lib.Context.get(x.device)
    fused = m(x).cpu().numpy()              # default: the split-bf16 kernel for the reference's [B,F,T] storage (F % 4 == 0), else exact fp32
    try:
        ctx.set_option("cnn1d_fused", 2)
        fused32 = m(x).cpu().numpy()        # always the exact-fp32 matrix-core kernel
        ctx.set_option("cnn1d_fused", 0)
        three = m(x).cpu().numpy()
    finally:
        ctx.set_option("cnn1d_fused", 1)
    want = O.cnn1d_forward(sd, (stored.numpy().swapaxes(1, 2) if layout == "bft_view" else stored.numpy()))
    np.testing.assert_allclose(fused, want, atol=TOL_F32, rtol=0)
    np.testing.assert_allclose(fused32, want, atol=TOL_F32, rtol=0)
    np.testing.assert_allclose(fused32, three, atol=2e-5, rtol=0)          # fp32 fma chains on both sides
    np.testing.assert_allclose(fused, three, atol=TOL_F32, rtol=0)         # hi + lo bf16 operands: 2^-17 per product
    print(f"[cnn1d fused {B},{T},{F},{layout}] max |x3 - oracle| {np.abs(fused - want).max():.2e}, |fp32 fused - oracle| {np.abs(fused32 - want).max():.2e}")
    assert np.isfinite(fused).all()


def test_cnn1d_fused_is_one_launch_and_batch_independent(golden):
    """One timing slot (4) fires per forward on the fused path, none of the per-layer slots; an utterance's logit does not depend
    on its neighbours in the batch (bit for bit: one workgroup per utterance)."""
    from dfa_amd import _lib
    sd, _ = golden("cnn1d_eval")
    model = _model(sd)
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(7, 180, 321, generator=g) * 3.2).to("cuda").transpose(1, 2)
    ctx = _lib.Context.get(x.device)
    ctx.timing_reset()
    ctx.timing(True)
    full = model(x)
    ctx.timing(False)
    torch.cuda.synchronize()
    counts = [ctx.timing_read(s)[1] for s in (4, 5, 6, 7)]
    ctx.timing_reset()
    assert counts == [1, 0, 0, 0], counts
    for i in (0, 3, 6):
        assert torch.equal(model(x[i:i + 1]), full[i:i + 1])


@pytest.mark.parametrize("B,T,F", [(3, 321, 180), (2, 64, 180), (4, 100, 44), (2, 384, 180), (2, 5, 180), (3, 33, 8), (2, 3, 4)])
def test_cnn1d_training_convolutions_on_matrix_cores_match_vector_twin(B, T, F):
    """Round 3: the five convolutions of a CNN1D training step (3 forward, 2 data gradients; src/train.py:71-76 through
    src/model_cnn1d.py:17-34) run on `conv1d_x3_kernel` (csrc/cnn1d_fused_x3.hip) when the tensors are in the stored layout:
    context option cnn1d_train_x3 = 1 -- every fp32 operand as three bf16 terms (its 24-bit mantissa exactly), six MFMAs per
    product -- must reproduce the fp32 VALU kernels (option 0) to fp32 rounding: logits 1e-6, gradients 1e-5 of their scale.
    Option 3 (two terms, bf16x3) is the opt-in fast form: logits within 1e-4, gradients limited by ReLU flips (a 1e-5
    perturbation flips ~1e-5 of the masks; relative L2 ~ 3e-3 .. 1e-2 at these batch sizes)."""
    from dfa_amd import _lib
    from dfa_amd.model_cnn1d import CNN1D
    g = torch.Generator().manual_seed(B * 1000 + T + F)
    x = (torch.randn(B, F, T, generator=g) * 3.2).to("cuda").transpose(1, 2)
    y = (torch.rand(B, generator=g) > 0.5).float().to("cuda")
    ctx = _lib.Context.get(x.device)
    res = {}
    try:
        for arm in (0, 1, 2, 3):
            ctx.set_option("cnn1d_train_x3", arm)
            torch.manual_seed(0)
            m = CNN1D(in_features=F, dropout=0.0).to("cuda").train()
            logits = m(x).squeeze(-1)
            torch.nn.BCEWithLogitsLoss()(logits, y).backward()
            res[arm] = (logits.detach().clone(), {n: p.grad.clone() for n, p in m.named_parameters()})
    finally:
        ctx.set_option("cnn1d_train_x3", 1)
    l0, g0 = res[0]
    wscale = {"conv.0.bias": "conv.0.weight", "conv.4.bias": "conv.4.weight", "conv.8.bias": "conv.8.weight"}   # conv biases in front of a BatchNorm: gradient = 0 up to rounding
    for arm, ltol, gtol in ((1, 2e-6, 1e-5), (2, 2e-6, 1e-5), (3, 1e-4, 5e-2 if B * T >= 1000 else 2e-1)):   # (a flip is 1 / (B T) of a channel's sum)
        l, gr = res[arm]
        assert float((l - l0).abs().max()) <= ltol * max(1.0, float(l0.abs().max())), (arm, float((l - l0).abs().max()))
        for n in g0:
            scale = float(g0[wscale.get(n, n)].abs().max())
            assert float((gr[n] - g0[n]).abs().max()) <= gtol * scale + 1e-12, (arm, n, float((gr[n] - g0[n]).abs().max()) / scale)
    assert torch.equal(res[1][0], res[2][0])        # one or two channel tiles per workgroup: the same sums
    for n in g0:
        assert torch.equal(res[1][1][n], res[2][1][n]), n
