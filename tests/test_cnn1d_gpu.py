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
    logits_c = model(stored.transpose(1, 2).contiguous())
    np.testing.assert_allclose(logits_c.cpu().numpy(), logits.cpu().numpy(), atol=1e-6, rtol=0)


def test_cnn1d_layers_match_golden(golden):
    sd, g = golden("cnn1d_eval")
    model = _model(sd)
    x = torch.from_numpy(g["t16.x_stored"]).to("cuda").transpose(1, 2)
    model(x)
    from dfa_amd import _lib
    ws = _lib.Context.get(x.device)._ws
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
