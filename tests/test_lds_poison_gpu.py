"""Stale-LDS robustness (GPU): LDS is not cleared between workgroups, and several kernels deliberately compute on lanes whose
operands lie past what the workgroup staged (the two spare columns of a 30-column strip, rows past the image) and mask those lanes
afterwards.  The masks must be selects: a 0/1 multiplier turns a NaN bit pattern left behind by an earlier kernel into a NaN
statistic once in a few thousand launches.  The "poison_lds" option of the C ABI fills all 160 KB of every CU's LDS with a chosen
16-bit pattern; every path must then reproduce its un-poisoned results bit for bit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PATTERNS = [0xFFFF, 0x7FC0, 0x7F80]     # NaN in either width / bf16 quiet NaN / bf16 +Inf


def _ctx():
    from dfa_amd import _lib
    return _lib.Context.get(torch.device("cuda"))


def _poison(pattern):
    _ctx().set_option("poison_lds", pattern)


def _stored(B, seed, T=321, F=180, dtype=torch.float32):
    gen = torch.Generator().manual_seed(seed)
    return (torch.randn(B, F, T, generator=gen) * 3.2 - 0.07).to("cuda").to(dtype).transpose(1, 2)


@pytest.mark.parametrize("prec", ["bf16", "bf16x3", "fp32"])
def test_cnn2d_eval_forward_ignores_stale_lds(prec):
    from dfa_amd.model import CNN2D
    torch.manual_seed(3)
    m = CNN2D(in_features=180, precision=prec).to("cuda")
    with torch.no_grad():                       # non-trivial BatchNorm state
        for i in m._BN_IDX:
            m.conv[i].running_mean.normal_(0, 0.3)
            m.conv[i].running_var.uniform_(0.5, 2.0)
    m = m.eval()
    for B, T, F in ((24, 321, 180), (3, 33, 47)):      # full-width strips and a ragged last strip
        x = _stored(B, 11, T, F, torch.bfloat16 if prec == "bf16" else torch.float32)
        want = m(x).clone() if F == 180 else None
        if F != 180:
            m2 = CNN2D(in_features=F, precision=prec).to("cuda").eval()
            want = m2(x).clone()
        for pat in PATTERNS:
            _poison(pat)
            got = (m if F == 180 else m2)(x)
            assert torch.isfinite(got).all()
            assert torch.equal(got, want), (prec, B, T, F, hex(pat))


@pytest.mark.parametrize("prec,B", [("bf16", 64), ("fp32", 16)])
def test_cnn2d_training_steps_ignore_stale_lds(prec, B):
    """Three optimisation steps (dropout on, fixed key) with LDS poisoned before every step equal the clean run bit for bit.
    (The failure this guards against was seen while trying 30-column strips for block 2's training forward: its two spare lanes
    read past the LDS ring, and a 0/1 multiplier summed 0 * NaN into the BatchNorm statistics every few thousand launches.)"""
    from dfa_amd.model import CNN2D
    from dfa_amd.training.train_step import NativeTrainer
    x = _stored(B, 31, dtype=torch.bfloat16 if prec == "bf16" else torch.float32)
    y = (torch.rand(B, generator=torch.Generator().manual_seed(2)) > 0.5).float().to("cuda")

    def run(pattern, ws_byte=None):
        torch.manual_seed(5)
        model = CNN2D(in_features=180, dropout=0.2, precision=prec).to("cuda")
        model._drop_seed = 99
        if ws_byte is not None:       # the workspace the C ABI is handed is uninitialised memory: make it hostile
            n = _ctx().lib.dfa_cnn2d_train_workspace_bytes(_ctx().handle, B, 321, 180, _lib.PRECISIONS[prec])
            model._train_ws = torch.full((int(n),), ws_byte, dtype=torch.uint8, device="cuda")
        tr = NativeTrainer(model, lr=1e-3, label_smoothing=0.05)
        losses = []
        for _ in range(3):
            if pattern is not None:
                _poison(pattern)
            losses.append(tr.step(x, y).clone())
        return torch.cat(losses), tr.flat_g.clone(), tr.flat_p.clone(), model.conv[6].running_var.clone()

    from dfa_amd import _lib
    want = run(None)
    assert all(torch.isfinite(t).all() for t in want)
    for pat in PATTERNS:
        got = run(pat)
        for u, v in zip(got, want):
            assert torch.equal(u, v), hex(pat)
    for ws_byte in (0xFF, 0x7F):      # workspace pre-filled with NaN patterns (bf16 0xffff / 0x7f7f = 3.4e38, fp32 NaN / 3.4e38)
        got = run(None, ws_byte)
        for u, v in zip(got, want):
            assert torch.equal(u, v), hex(ws_byte)


def test_cnn2d_long_training_run_stays_finite():
    """Forty full-size bf16 steps: losses, gradients and parameters stay finite (a NaN statistic anywhere turns the parameters
    NaN within a step, and the ReLUs then hide it behind a loss of exactly ln 2)."""
    from dfa_amd.model import CNN2D
    from dfa_amd.training.train_step import NativeTrainer
    x = _stored(256, 1, dtype=torch.bfloat16)
    y = (torch.rand(256, generator=torch.Generator().manual_seed(1)) > 0.5).float().to("cuda")
    torch.manual_seed(0)
    model = CNN2D(in_features=180, dropout=0.2, precision="bf16").to("cuda")
    tr = NativeTrainer(model, lr=1e-6, label_smoothing=0.05)
    for step in range(40):
        if step % 4 == 0:
            _poison(PATTERNS[(step // 4) % 3])
        loss = tr.step(x, y)
        if step % 4 == 3:
            assert torch.isfinite(loss).all() and torch.isfinite(tr.flat_g).all() and torch.isfinite(tr.flat_p).all(), step
    for i in model._BN_IDX:
        assert torch.isfinite(model.conv[i].running_mean).all() and torch.isfinite(model.conv[i].running_var).all()


def test_cnn1d_and_cae_ignore_stale_lds():
    from dfa_amd.model_cnn1d import CNN1D
    from dfa_amd.model_cae import ConvAutoencoder
    torch.manual_seed(7)
    x = _stored(16, 5)
    m1 = CNN1D(in_features=180, dropout=0.0).to("cuda").eval()
    want1 = m1(x).clone()
    xc = torch.randn(8, 64, 180, generator=torch.Generator().manual_seed(9)).to("cuda")
    caes = {p: ConvAutoencoder(precision=p).to("cuda").eval() for p in ("fp32", "bf16")}
    wantc = {p: tuple(t.clone() for t in m(xc)) for p, m in caes.items()}
    for pat in PATTERNS:
        _poison(pat)
        assert torch.equal(m1(x), want1), hex(pat)
        for p, m in caes.items():
            _poison(pat)
            got = m(xc)
            assert all(torch.equal(a, b) for a, b in zip(got, wantc[p])), (p, hex(pat))


@pytest.mark.parametrize("which", ["cnn1d", "cae_fp32", "cae_bf16"])
def test_cnn1d_and_cae_training_ignore_stale_lds(which):
    from dfa_amd.model_cnn1d import CNN1D
    from dfa_amd.model_cae import ConvAutoencoder

    def run(pattern, ws_byte=None):
        torch.manual_seed(13)
        if which == "cnn1d":
            m = CNN1D(in_features=180, dropout=0.0).to("cuda").train()
            x = _stored(16, 5)
            y = (torch.rand(16, generator=torch.Generator().manual_seed(3)) > 0.5).float().to("cuda")
        else:
            m = ConvAutoencoder(precision=which[4:]).to("cuda").train()
            x = torch.randn(4, 64, 180, generator=torch.Generator().manual_seed(9)).to("cuda")
        if ws_byte is not None:       # a hostile (NaN-pattern) workspace, larger than any of these shapes needs
            m._train_ws = torch.full((192 << 20,), ws_byte, dtype=torch.uint8, device="cuda")
        if pattern is not None:
            _poison(pattern)
        if which == "cnn1d":
            loss = torch.nn.BCEWithLogitsLoss()(m(x).squeeze(-1), y)
        else:
            loss = torch.nn.MSELoss()(m(x)[0], x)
        loss.backward()
        return [loss.detach().clone()] + [p.grad.clone() for p in m.parameters()]

    want = run(None)
    assert all(torch.isfinite(t).all() for t in want)
    for pat in PATTERNS:
        for u, v in zip(run(pat), want):
            assert torch.equal(u, v), hex(pat)
    for u, v in zip(run(None, 0xFF), want):
        assert torch.equal(u, v), "workspace"
