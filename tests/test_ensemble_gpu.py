"""GPU test of the three-model ensemble path (BASELINE configs[4]): CNN2D + CNN1D + CAE score the same resident batch,
scores are fused with the reference formulas; everything is compared with the CPU oracle run on the same inputs."""
import numpy as np
import pytest
import torch

from oracle import dfa_oracle as O

pytestmark = pytest.mark.gpu


def _load(cls, sd, **kw):
    m = cls(**kw)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.to("cuda").eval()


def test_hybrid_ensemble_matches_oracle(golden):
    from dfa_amd.dataset_cae import FeatureNormalizer
    from dfa_amd.hybrid_ensemble import hybrid_report, score_models
    from dfa_amd.model import CNN2D
    from dfa_amd.model_cae import ConvAutoencoder
    from dfa_amd.model_cnn1d import CNN1D
    sd2, _ = golden("cnn2d_eval")
    sd1, _ = golden("cnn1d_eval")
    sdc, _ = golden("cae_eval")
    n = 24
    g = torch.Generator().manual_seed(21)
    labels = (torch.rand(n, generator=g) > 0.5).long()
    pattern = torch.outer(torch.sin(torch.arange(180) / 9.0), torch.cos(torch.arange(321) / 31.0))
    stored = torch.stack([torch.randn(180, 321, generator=g) * 3.2 - 0.07 + 0.5 * labels[i] * pattern for i in range(n)])
    norm = FeatureNormalizer().fit([s.transpose(0, 1) for s in stored[labels == 1]])
    models = (_load(CNN2D, sd2), _load(CNN1D, sd1), _load(ConvAutoencoder, sdc))
    whole = score_models(stored, *models, normalizer=norm, batch_size=7, device="cuda")
    # two "ranks" scoring contiguous shards reproduce the unsharded vectors (no cross-sample op anywhere)
    parts = [score_models(stored, *models, normalizer=norm, batch_size=5, device="cuda", rank=r, world=2) for r in (0, 1)]
    for k in ("cnn2d", "cnn1d", "cae"):
        np.testing.assert_allclose(np.concatenate([p[k] for p in parts]), whole[k], rtol=1e-6, atol=1e-7)
    x = stored.numpy().swapaxes(1, 2)
    want2 = O.sigmoid(O.cnn2d_forward(sd2, x)[:, 0])
    want1 = O.sigmoid(O.cnn1d_forward(sd1, x)[:, 0])
    xz = O.normalizer_transform(x, norm.mean.numpy(), norm.std.numpy())
    recon, _ = O.cae_forward(sdc, xz)
    wantc = O.per_sample_mse(recon, xz)
    np.testing.assert_allclose(whole["cnn2d"], want2, atol=1e-4)
    np.testing.assert_allclose(whole["cnn1d"], want1, atol=1e-4)
    np.testing.assert_allclose(whole["cae"], wantc, rtol=2e-5)
    rep = hybrid_report(whole["cnn2d"], whole["cae"], labels.tolist())
    table, best_eer, best_alpha = O.hybrid_alpha_sweep(want2, wantc.astype(np.float64), labels.tolist())
    assert [e for _, e in rep["table"]] == [e for _, e in table] and rep["best_eer"] == best_eer
