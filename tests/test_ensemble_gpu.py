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


def test_predict_hybrid_and_ensemble_clis(golden, tmp_path):
    """Round 3: the CLI mirrors of src/predict_hybrid.py (fixed-alpha prediction file + comparison with an existing submission)
    and src/ensemble.py (mean of sigmoids over arch:path checkpoints) on the HIP path, end to end on files in the reference's
    schema, against the oracle's scores fused with the reference's formulas."""
    import pandas as pd
    from dfa_amd import ensemble as ens_cli, predict_hybrid as ph_cli
    from dfa_amd.dataset_cae import FeatureNormalizer
    sd2, _ = golden("cnn2d_eval")
    sd1, _ = golden("cnn1d_eval")
    sdc, _ = golden("cae_eval")
    n = 20
    g = torch.Generator().manual_seed(33)
    labels = (torch.rand(n, generator=g) > 0.5).long()
    labels[0], labels[1] = 0, 1
    pattern = torch.outer(torch.sin(torch.arange(180) / 7.0), torch.cos(torch.arange(321) / 29.0))
    stored = torch.stack([torch.randn(180, 321, generator=g) * 3.2 - 0.07 + 0.6 * labels[i] * pattern for i in range(n)])
    uttids = [f"utt_{i:04d}" for i in range(n)]
    fpath, lpath = str(tmp_path / "features.pkl"), str(tmp_path / "labels.pkl")
    pd.DataFrame({"uttid": uttids, "features": [stored[i].clone() for i in range(n)]}).to_pickle(fpath)
    pd.DataFrame({"uttid": uttids, "label": labels.tolist()}).to_pickle(lpath)
    norm = FeatureNormalizer().fit([s.transpose(0, 1) for s in stored[labels == 1]])
    npath = str(tmp_path / "normalizer.pt")
    norm.save(npath)
    ck = {}
    for name, sd, wrap in (("cnn2d", sd2, True), ("cnn1d", sd1, False), ("cae", sdc, True)):
        state = {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
        ck[name] = str(tmp_path / f"{name}.pt")
        torch.save({"model_state": state} if wrap else state, ck[name])           # both checkpoint forms (src/predict.py:82-85)
    # oracle scores
    x = stored.numpy().swapaxes(1, 2)
    want2 = O.sigmoid(O.cnn2d_forward(sd2, x)[:, 0])
    want1 = O.sigmoid(O.cnn1d_forward(sd1, x)[:, 0])
    xz = O.normalizer_transform(x, norm.mean.numpy(), norm.std.numpy())
    recon, _ = O.cae_forward(sdc, xz)
    wantc = O.per_sample_mse(recon, xz).astype(np.float64)

    def n01(v):
        return np.zeros_like(v) if v.max() - v.min() < 1e-12 else (v - v.min()) / (v.max() - v.min())
    # --- predict_hybrid: alpha * norm(sup) + (1 - alpha) * norm(cae)   (src/predict_hybrid.py:81-85,149-151)
    out = str(tmp_path / "prediction_hybrid.pkl")
    old = str(tmp_path / "old_submission.pkl")
    pd.to_pickle({"student_id": "x", "predictions": pd.DataFrame({"uttid": uttids, "predictions": want2})}, old)
    pred = ph_cli.main(["--sup-checkpoint", ck["cnn2d"], "--cae-checkpoint", ck["cae"], "--cae-normalizer", npath,
                        "--test-features", fpath, "--alpha", "0.7", "--out", out, "--existing-submission", old, "--batch-size", "6"])
    got = pd.read_pickle(out)
    assert list(got.columns) == ["uttid", "predictions"] and got["predictions"].dtype == np.float64
    assert got["uttid"].tolist() == uttids and pred["predictions"].equals(got["predictions"])
    np.testing.assert_allclose(got["predictions"].values, 0.7 * n01(want2.astype(np.float64)) + 0.3 * n01(wantc), atol=2e-4)
    # --- ensemble: mean of sigmoids over arch:path members, per-member and ensemble EER   (src/ensemble.py:100-131)
    eout = str(tmp_path / "prediction_ens.pkl")
    res = ens_cli.main(["--checkpoints", f"cnn2d:{ck['cnn2d']}", f"cnn1d:{ck['cnn1d']}", "--dev-features", fpath, "--dev-labels", lpath,
                        "--batch-size", "8", "--out", eout])
    want_ens = np.mean([want2, want1], axis=0)
    np.testing.assert_allclose(res["scores"], want_ens, atol=1e-4)
    assert res["eer"] == O.calculate_eer(want_ens.tolist(), labels.tolist())[0]
    assert [m[0] for m in res["members"]] == ["cnn2d", "cnn1d"]
    assert res["members"][0][2] == O.calculate_eer(want2.tolist(), labels.tolist())[0]
    assert pd.read_pickle(eout)["uttid"].tolist() == uttids
    with pytest.raises(ValueError):
        ens_cli.main(["--checkpoints", f"mlp:{ck['cnn2d']}", "--dev-features", fpath, "--dev-labels", lpath])
