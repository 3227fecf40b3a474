"""Round-2 GPU parity tests (all through the C ABI):
  * the bf16 storage mode (the headline configuration) against the ROUNDING-FAITHFUL oracle (emulate="bf16"), at
    accumulation-order tolerance instead of the loose bf16 bound;
  * the N=2000 EER-parity set of SURVEY 8(d) against the reference's own predictions / EER (tests/golden/eer2000.npz);
  * FusedAugment against the reference-generated augmentation fixtures;
  * a checkpoint written by the reference's save_checkpoint, loaded and run on the GPU;
  * flat-file ingest -> FlatBatcher -> kernels against the pickle path."""
import os
import random

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import GOLDEN
from oracle import dfa_oracle as O
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu

TOL_F32 = 1e-4        # north_star: logits within 1e-4 in fp32
# bf16 storage mode vs the rounding-faithful oracle: what remains is fp32 accumulation order plus the rare bf16
# re-rounding of an activation whose fp32 value sits within an ulp of a rounding boundary.  Measured on MI355X
# (tools/gpu_bf16_emu_probe.py, round 2): max |dlogit| 1.9e-5 / 1.4e-4 / 2.6e-4 on the golden T = 321 / 64 / 7 cases at
# |logit| ~ 3, where the fp32 reference sits 1.7e-3 .. 9e-3 away -> bound 1e-3 relative to max(1, |logit|).
TOL_BF16_EMU_REL = 1e-3


def _model_from_sd(sd, precision="fp32"):
    from dfa_amd.model import CNN2D
    m = CNN2D(precision=precision)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.to("cuda").eval()


def _a2_view(x, B, T, F):
    """bf16 a2 [B, T/4, F, 64] left in the workspace by the last bf16-mode forward (layout: api.hip plan_cnn2d)."""
    from dfa_amd import _lib
    ws = _lib.Context.get(x.device)._ws
    H1, H2 = T // 2, T // 4
    off2 = (B * H1 * F * 32 * 2 + 255) // 256 * 256
    return ws[off2:off2 + B * H2 * F * 64 * 2].view(torch.bfloat16).view(B, H2, F, 64).float().cpu().numpy()


@pytest.mark.parametrize("tag", ["t321", "t64", "t7"])
def test_cnn2d_bf16_mode_matches_rounding_faithful_oracle(golden, tag):
    sd, g = golden("cnn2d_eval")
    model = _model_from_sd(sd, "bf16")
    xs = g[f"{tag}.x_stored"]
    x = torch.from_numpy(xs).to("cuda").transpose(1, 2)
    B, T, F = x.shape
    logits, emb = model(x, return_embedding=True)
    a2 = _a2_view(x, B, T, F)
    want, inter = O.cnn2d_forward(sd, np.swapaxes(xs, 1, 2), return_intermediates=True, emulate="bf16")
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(logits.cpu().numpy(), want, atol=TOL_BF16_EMU_REL * scale, rtol=0)
    np.testing.assert_allclose(emb.cpu().numpy(), inter["embedding"], atol=2e-3 * max(1.0, float(np.abs(inter["embedding"]).max())), rtol=0)
    # a2 is stored in bf16 on both sides: elements are EQUAL except where an fp32 sum sat on a rounding boundary; then they
    # differ by one bf16 ulp (2^-8 relative), plus -- where a re-rounded a1 element feeds a sum with cancellation -- a
    # little more in absolute terms (2e-3 against an rms of 0.34).  An indexing bug in a halo column (differences of the
    # order of the values, in whole columns) cannot hide here.
    ref_a2 = inter["a2"].transpose(0, 2, 3, 1)
    diff = np.abs(a2 - ref_a2)
    assert float((diff > 0).mean()) < 0.005, float((diff > 0).mean())
    assert np.all(diff <= 2.0 ** -7 * np.abs(ref_a2) + 2e-3), float((diff - 2.0 ** -7 * np.abs(ref_a2)).max())
    # and the fp32 reference sits at bf16 distance (sanity: the oracle emulates, it does not replace, the reference)
    assert np.abs(logits.cpu().numpy() - g[f"{tag}.logits"]).max() < 0.05 * scale


@pytest.mark.parametrize("B,T", [(5, 33), (2, 4), (3, 130)])
def test_cnn2d_bf16_mode_matches_emulated_oracle_random_shapes(golden, B, T):
    sd, _ = golden("cnn2d_eval")
    model = _model_from_sd(sd, "bf16")
    g = torch.Generator().manual_seed(900 + B * 1000 + T)
    stored = torch.randn(B, 180, T, generator=g) * 3.2 - 0.07
    want = O.cnn2d_forward(sd, stored.numpy().swapaxes(1, 2), emulate="bf16")
    for xin in (stored.to("cuda"), stored.to("cuda").to(torch.bfloat16)):        # fp32 features are rounded on load
        got = model(xin.transpose(1, 2)).cpu().numpy()
        np.testing.assert_allclose(got, want, atol=TOL_BF16_EMU_REL * max(1.0, float(np.abs(want).max())), rtol=0)


def test_cnn2d_bf16_full_batch_matches_emulated_oracle(golden):
    """BASELINE configs[1] shape [256,321,180] in the headline mode: a sample of utterances against the torch twin of
    the rounding-faithful oracle, and batch independence at that size."""
    sd, _ = golden("cnn2d_eval")
    model = _model_from_sd(sd, "bf16")
    g = torch.Generator().manual_seed(17)
    stored = torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07
    x = stored.to("cuda").to(torch.bfloat16).transpose(1, 2)
    full = model(x).cpu()
    idx = [0, 1, 77, 200, 255]
    want = R.cnn2d_forward_emulated(sd, stored[idx].transpose(1, 2), "bf16")
    np.testing.assert_allclose(full[idx].numpy(), want.numpy(), atol=TOL_BF16_EMU_REL * max(1.0, float(want.abs().max())), rtol=0)
    for i in (3, 254):
        assert torch.equal(model(x[i:i + 1]).cpu(), full[i:i + 1])


# ------------------------------------------------------------------------------------------------ N = 2000 EER set
@pytest.fixture(scope="module")
def eer_files(tmp_path_factory):
    import eer_set
    z = np.load(os.path.join(GOLDEN, "eer2000.npz"))
    feats, labels, uttids = eer_set.make_eer_set(z["pattern"])
    assert np.array_equal(labels.numpy(), z["labels"])
    assert abs(eer_set.checksum(feats) - float(z["checksum"])) <= 1e-9 * abs(float(z["checksum"])), \
        "the torch CPU generator no longer reproduces the committed feature set"
    td = tmp_path_factory.mktemp("eer2000")
    fdf, ldf = eer_set.to_frames(feats, labels, uttids)
    fp, lp, ck = str(td / "features.pkl"), str(td / "labels.pkl"), str(td / "cnn2d.pt")
    fdf.to_pickle(fp)
    ldf.to_pickle(lp)
    from conftest import load_golden
    sd, _ = load_golden("cnn2d_eval")
    torch.save({"model_state": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}}, ck)
    return z, td, fp, lp, ck, labels.numpy()


def test_eer2000_fp32_predictions_and_eer_identical_to_reference(eer_files):
    """features.pkl -> dfa_amd.predict -> prediction.pkl -> scorer on the N=2000 set, against what the reference's own
    `src/predict.py` + `scripts/evaluation.py` produced on the same file: every score within 1e-4, EER and threshold
    index IDENTICAL for sigmoid and raw-logit scores (north_star: identical EER via scripts/evaluation.py)."""
    from dfa_amd import evaluation, predict
    z, td, fp, lp, ck, labels = eer_files
    for tag, flag in (("sigmoid", []), ("logits", ["--no-apply-sigmoid"])):
        out = str(td / f"prediction_{tag}.pkl")
        predict.main(["--features", fp, "--checkpoint", ck, "--model", "cnn2d", "--out", out, "--batch-size", "32"] + flag)
        got = pd.read_pickle(out)
        ref = z[f"{tag}.predictions"]
        assert got["predictions"].dtype == np.float64 and len(got) == 2000
        np.testing.assert_allclose(got["predictions"].values, ref, atol=TOL_F32, rtol=0)
        res = evaluation.score_prediction_file(out, lp)
        want_eer, want_thr = z[f"{tag}.eer"]
        assert 0.0 < want_eer < 0.05
        assert res["eer"] == want_eer, (tag, res["eer"], want_eer)
        # the threshold is the score at the same rank: same utterance, value within the logit tolerance
        assert abs(res["threshold"] - want_thr) <= TOL_F32, (tag, res["threshold"], want_thr)


def test_eer2000_bf16_mode_eer_within_one_hundredth_percent(eer_files):
    """The headline (bf16 storage) mode on the same set: raw-logit EER vs the reference's, |dEER| <= 0.01 % abs
    (north_star), and the scores against the rounding-faithful oracle on a sample."""
    from dfa_amd import evaluation, predict
    from conftest import load_golden
    z, td, fp, lp, ck, labels = eer_files
    out = str(td / "prediction_bf16.pkl")
    predict.main(["--features", fp, "--checkpoint", ck, "--model", "cnn2d", "--out", out, "--batch-size", "256",
                  "--no-apply-sigmoid", "--precision", "bf16"])
    got = pd.read_pickle(out)["predictions"].values
    ref = z["logits.predictions"]
    assert np.abs(got - ref).max() < 0.05 * max(1.0, np.abs(ref).max())
    eer16 = evaluation.score_prediction_file(out, lp)["eer"]
    eer32 = float(z["logits.eer"][0])
    swaps = int(np.sum(labels[np.argsort(got, kind="stable")] != labels[np.argsort(ref, kind="stable")]))
    print(f"bf16 EER {eer16:.6f} vs reference {eer32:.6f} (|d| = {abs(eer16 - eer32):.2e}); rank positions whose label "
          f"differs: {swaps} of 2000; max |dlogit| {np.abs(got - ref).max():.3e}")
    assert abs(eer16 - eer32) <= 1e-4, (eer16, eer32, swaps)
    sd, _ = load_golden("cnn2d_eval")
    feats = torch.stack(list(pd.read_pickle(fp)["features"].iloc[:8]))
    want = R.cnn2d_forward_emulated(sd, feats.transpose(1, 2), "bf16").squeeze(-1).numpy()
    np.testing.assert_allclose(got[:8], want, atol=TOL_BF16_EMU_REL * max(1.0, float(np.abs(want).max())), rtol=0)


# ------------------------------------------------------------------------------------------------ augmentation
def test_fused_augment_matches_reference_fixtures():
    """dfa_augment_batch (one HIP pass) under the reference's seeds == the output the reference's own
    spec_augment -> time_shift -> channel_drop chain produced (tests/golden/augment.npz), for both feature layouts and
    for fp32 / bf16 batches (every op is a select or an exact multiply, so bf16 commutes with the rounding)."""
    from dfa_amd.augmentation import FusedAugment
    from test_fixtures_r2 import aug_cases
    for tag, cfg, g in aug_cases():
        seed = int(g["seed"])
        for layout in ("btf", "bft_view"):
            for dtype in (torch.float32, torch.bfloat16):
                x = torch.from_numpy(g["x"])
                want = torch.from_numpy(g["y"]).to(dtype)
                if layout == "bft_view":
                    xd = x.transpose(1, 2).contiguous().to("cuda", dtype=dtype).transpose(1, 2)
                else:
                    xd = x.to("cuda", dtype=dtype)
                random.seed(seed); torch.manual_seed(seed)
                got = FusedAugment(gaussian_jitter=False, rng_device="cpu", **cfg)(xd)
                assert got.shape == xd.shape and got.dtype == dtype
                assert torch.equal(got.cpu(), want), (tag, layout, dtype)


# ------------------------------------------------------------------------------------------------ checkpoint / ingest
def test_reference_written_checkpoint_runs_on_gpu():
    """ref_cnn2d_checkpoint.pt was written by the reference's save_checkpoint; dfa_amd.predict loads it and the HIP
    forward reproduces the logits the reference model gave before saving."""
    from dfa_amd.predict import build_model, load_weights
    z = np.load(os.path.join(GOLDEN, "ref_cnn2d_checkpoint_io.npz"))
    model = build_model("cnn2d", 180, 0.3, "fp32").to("cuda")
    load_weights(model, os.path.join(GOLDEN, "ref_cnn2d_checkpoint.pt"), "cuda")
    model.eval()
    got = model(torch.from_numpy(z["x_stored"]).to("cuda").transpose(1, 2)).cpu().numpy()
    np.testing.assert_allclose(got, z["logits"], atol=TOL_F32, rtol=0)


def test_flat_ingest_to_kernels_equals_pickle_path(tmp_path, golden):
    """ingest.convert -> FlatFeatures (memmap) -> FlatBatcher (pinned, double-buffered H2D) -> kernels gives
    the same scores as the pickle reader, bit for bit, in fp32 and in bf16 storage."""
    from dfa_amd import ingest
    from dfa_amd.predict import predict_scores
    sd, _ = golden("cnn2d_eval")
    n = 37
    g = torch.Generator().manual_seed(3)
    feats = [torch.randn(180, 321, generator=g) * 3.2 - 0.07 for _ in range(n)]
    uttids = [f"u{i:04d}" for i in range(n)]
    fp = str(tmp_path / "features.pkl")
    pd.DataFrame({"uttid": uttids, "features": feats}).to_pickle(fp)
    stack = torch.stack(list(pd.read_pickle(fp)["features"]))
    for dtype, prec in (("fp32", "fp32"), ("bf16", "bf16")):
        model = _model_from_sd(sd, prec)
        ingest.convert(fp, str(tmp_path / f"flat_{dtype}"), None, dtype=dtype)
        ff = ingest.FlatFeatures(str(tmp_path / f"flat_{dtype}"))
        assert ff.uttids == uttids and len(ff) == n
        a = predict_scores(model, ff.tensor(), batch_size=16, apply_sigmoid=False)
        b = predict_scores(model, stack if dtype == "fp32" else stack.to(torch.bfloat16), batch_size=16, apply_sigmoid=False)
        assert torch.equal(a, b), dtype
        want = R.cnn2d_forward(sd, stack[:4].transpose(1, 2)).squeeze(-1)
        tol = TOL_F32 if prec == "fp32" else 0.05 * float(want.abs().max())
        np.testing.assert_allclose(a[:4].cpu().numpy(), want.numpy(), atol=tol, rtol=0)


# ------------------------------------------------------------------------------------------------ ctx weight-slot ownership
def test_two_models_of_one_class_interleaved(golden):
    """A dfa_ctx has one weight slot per model class: two live CNN2D (and CNN1D / CAE) instances with different weights
    used alternately (A, B, A) must each get their own weights every time, in eval and in train mode."""
    from dfa_amd.model import CNN2D
    from dfa_amd.model_cnn1d import CNN1D
    sd, g = golden("cnn2d_eval")
    xs = g["t64.x_stored"]
    x = torch.from_numpy(xs).to("cuda").transpose(1, 2)
    a = _model_from_sd(sd)
    torch.manual_seed(123)
    b = CNN2D().to("cuda").eval()
    with torch.no_grad():
        b.classifier.weight.mul_(40.0)
    sdb = {k: v.cpu().numpy() for k, v in b.state_dict().items()}
    want_a, want_b = g["t64.logits"], O.cnn2d_forward(sdb, np.swapaxes(xs, 1, 2))
    assert np.abs(want_a - want_b).max() > 1e-2                                # the two models really differ
    for m, want in ((a, want_a), (b, want_b), (a, want_a), (a, want_a), (b, want_b)):
        np.testing.assert_allclose(m(x).cpu().numpy(), want, atol=TOL_F32, rtol=0)
    for prec in ("bf16", "fp32"):                                              # precision switches interleaved too
        a.set_precision(prec); b.set_precision(prec)
        la, lb, la2 = a(x), b(x), a(x)
        assert torch.equal(la, la2) and not torch.equal(la, lb)
    a.set_precision("fp32"); b.set_precision("fp32")
    # train mode: forward(A) -> forward(B) -> backward(A) must not silently use B's state
    a.train(); b.train()
    xt = x[:, :16].contiguous()
    la = a(xt)
    lb = b(xt)
    with pytest.raises(RuntimeError, match="ANOTHER cnn2d model|no longer the latest"):
        la.sum().backward()
    lb.sum().backward()                                                        # the latest forward is fine
    l1 = a(xt)
    l2 = a(xt)                                                                 # same model, two forwards: the first graph is stale
    with pytest.raises(RuntimeError, match="no longer the latest"):
        l1.sum().backward()
    l2.sum().backward()
    a.eval(); b.eval()
    np.testing.assert_allclose(b(x).cpu().numpy().shape, want_b.shape)
    # CNN1D shares the mechanism
    sd1, g1 = golden("cnn1d_eval")
    c = CNN1D(); c.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd1.items()}); c = c.to("cuda").eval()
    torch.manual_seed(5)
    d = CNN1D().to("cuda").eval()
    x1 = torch.from_numpy(g1["t64.x_stored"]).to("cuda").transpose(1, 2)
    wc = g1["t64.logits"]
    wd = O.cnn1d_forward({k: v.cpu().numpy() for k, v in d.state_dict().items()}, np.swapaxes(g1["t64.x_stored"], 1, 2))
    for m, want in ((c, wc), (d, wd), (c, wc)):
        np.testing.assert_allclose(m(x1).cpu().numpy(), want, atol=TOL_F32, rtol=0)


# ------------------------------------------------------------------------------------------------ bf16x3 (split-bf16) mode
@pytest.mark.parametrize("tag", ["t321", "t64", "t7", "t16"])
def test_cnn2d_bf16x3_matches_golden_at_fp32_tolerance(golden, tag):
    """The hi + lo split mode (three bf16 MFMAs per product) must meet the SAME bar as the exact-fp32 mode: the reference's
    logits within 1e-4, embeddings to fp32-level tolerance, strided and contiguous inputs alike."""
    sd, g = golden("cnn2d_eval")
    model = _model_from_sd(sd, "bf16x3")
    stored = torch.from_numpy(g[f"{tag}.x_stored"]).to("cuda")
    x = stored.transpose(1, 2)
    logits, emb = model(x, return_embedding=True)
    np.testing.assert_allclose(logits.cpu().numpy(), g[f"{tag}.logits"], atol=TOL_F32, rtol=0)
    if tag != "t16":
        np.testing.assert_allclose(emb.cpu().numpy(), g[f"{tag}.embedding"], atol=3e-5, rtol=3e-5)
    assert torch.equal(model(x.contiguous()), logits)


@pytest.mark.parametrize("B,T", [(1, 321), (5, 33), (2, 4), (3, 130), (2, 5)])
def test_cnn2d_bf16x3_matches_oracle_random_shapes(golden, B, T):
    sd, _ = golden("cnn2d_eval")
    model = _model_from_sd(sd, "bf16x3")
    g = torch.Generator().manual_seed(300 + B * 1000 + T)
    stored = torch.randn(B, 180, T, generator=g) * 3.2 - 0.07
    want, inter = O.cnn2d_forward(sd, stored.numpy().swapaxes(1, 2), return_intermediates=True)
    logits, emb = model(stored.to("cuda").transpose(1, 2), return_embedding=True)
    np.testing.assert_allclose(emb.cpu().numpy(), inter["embedding"], atol=3e-5, rtol=3e-5)
    np.testing.assert_allclose(logits.cpu().numpy(), want, atol=TOL_F32, rtol=0)


def test_cnn2d_bf16x3_ragged_widths_twins_and_batch_independence(golden):
    """F = 40 / 65 (ragged last strip), the asm-pipelined kernels against their compiler-scheduled twins (bit-identical),
    and every utterance of a [256,321,180] batch equal to its batch-of-one result."""
    from dfa_amd import _lib
    from dfa_amd.model import CNN2D
    ctx = _lib.Context.get(torch.device("cuda"))
    for F in (40, 65):
        torch.manual_seed(F)
        m = CNN2D(in_features=F, precision="bf16x3").to("cuda").eval()
        with torch.no_grad():
            m.classifier.weight.mul_(30.0)
        sdf = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
        x = torch.randn(2, 21, F)
        np.testing.assert_allclose(m(x.to("cuda")).cpu().numpy(), O.cnn2d_forward(sdf, x.numpy()), atol=TOL_F32, rtol=0)
    sd, g = golden("cnn2d_eval")
    model = _model_from_sd(sd, "bf16x3")
    gen = torch.Generator().manual_seed(7)
    stored = (torch.randn(256, 180, 321, generator=gen) * 3.2 - 0.07).to("cuda")
    x = stored.transpose(1, 2)
    try:
        ctx.set_option("lds_pipe", 0)
        l0, e0 = model(x, return_embedding=True)
        ctx.set_option("lds_pipe", 1)
        l1, e1 = model(x, return_embedding=True)
    finally:
        ctx.set_option("lds_pipe", 1)
    assert torch.equal(l0, l1) and torch.equal(e0, e1)
    for idx in (0, 100, 255):
        assert torch.equal(model(x[idx:idx + 1]), l1[idx:idx + 1])
    want = R.cnn2d_forward(sd, stored[:3].cpu().transpose(1, 2))
    np.testing.assert_allclose(l1[:3].cpu().numpy(), want.numpy(), atol=TOL_F32, rtol=0)
    # bf16 features are accepted too (they are exact in the hi plane)
    xb = stored[:4].to(torch.bfloat16)
    want_b = R.cnn2d_forward(sd, xb.float().cpu().transpose(1, 2))
    np.testing.assert_allclose(model(xb.transpose(1, 2)).cpu().numpy(), want_b.numpy(), atol=TOL_F32, rtol=0)


def test_eer2000_bf16x3_predictions_and_eer_identical_to_reference(eer_files):
    """The N=2000 set in the split mode: every score within 1e-4 of the reference's, EER identical (sigmoid and raw)."""
    from dfa_amd import evaluation, predict
    z, td, fp, lp, ck, labels = eer_files
    for tag, flag in (("sigmoid", []), ("logits", ["--no-apply-sigmoid"])):
        out = str(td / f"prediction_x3_{tag}.pkl")
        predict.main(["--features", fp, "--checkpoint", ck, "--model", "cnn2d", "--out", out, "--batch-size", "256",
                      "--precision", "bf16x3"] + flag)
        got = pd.read_pickle(out)
        np.testing.assert_allclose(got["predictions"].values, z[f"{tag}.predictions"], atol=TOL_F32, rtol=0)
        res = evaluation.score_prediction_file(out, lp)
        assert res["eer"] == float(z[f"{tag}.eer"][0]), (tag, res["eer"])


def test_bf16x3_is_eval_only(golden):
    sd, g = golden("cnn2d_eval")
    model = _model_from_sd(sd, "bf16x3").train()
    with pytest.raises(ValueError):
        model(torch.zeros(2, 16, 180, device="cuda"))


# ------------------------------------------------------------------------------------------------ small batches: time-axis split
@pytest.mark.parametrize("prec", ["bf16", "bf16x3"])
def test_time_axis_split_is_bit_invariant(golden, prec):
    """Small batches (the reference's predict.py default is 32) split the time axis over workgroups.  The time mean is always
    summed in canonical chunks added in chunk order, so logits and embeddings must be IDENTICAL bit for bit whether the
    axis is split automatically, not at all, or into any forced number of segments -- and for any batch size."""
    from dfa_amd import _lib
    sd, g = golden("cnn2d_eval")
    ctx = _lib.Context.get(torch.device("cuda"))
    model = _model_from_sd(sd, prec)
    gen = torch.Generator().manual_seed(41)
    try:
        for (B, T) in ((3, 321), (2, 130), (5, 37), (1, 700), (2, 9)):
            stored = torch.randn(B, 180, T, generator=gen) * 3.2 - 0.07
            x = stored.to("cuda").transpose(1, 2)
            ctx.set_option("time_split", 0)
            l0, e0 = model(x, return_embedding=True)
            for split in (-1, 2, 3, 5, 8):
                ctx.set_option("time_split", split)
                l1, e1 = model(x, return_embedding=True)
                assert torch.equal(e0, e1) and torch.equal(l0, l1), (prec, B, T, split, float((e0 - e1).abs().max()))
                l2 = model(x)                                   # without the embedding output
                assert torch.equal(l2, l0), (prec, B, T, split)
        # against the references once more, with the split on
        ctx.set_option("time_split", -1)
        x = torch.from_numpy(g["t321.x_stored"]).to("cuda").transpose(1, 2)
        got = model(x).cpu().numpy()
        if prec == "bf16x3":
            np.testing.assert_allclose(got, g["t321.logits"], atol=TOL_F32, rtol=0)
        else:
            want = O.cnn2d_forward(sd, np.swapaxes(g["t321.x_stored"], 1, 2), emulate="bf16")
            np.testing.assert_allclose(got, want, atol=TOL_BF16_EMU_REL * max(1.0, float(np.abs(want).max())), rtol=0)
    finally:
        ctx.set_option("time_split", -1)


# ------------------------------------------------------------------------------------------------ tracing hook
def test_roctx_ranges_do_not_change_results(golden):
    """DFA_ROCTX=1 makes every C-ABI forward / backward / optimiser call push a named roctx range (csrc/trace.hip; the marker
    library is dlopen'ed, no tool attached here so the ranges go nowhere).  The child process must load the marker library,
    run the forward and reproduce this process's logits bit for bit."""
    import subprocess, sys, os, tempfile
    sd, g = golden("cnn2d_eval")
    model = _model_from_sd(sd, "bf16")
    x = torch.from_numpy(g["t321.x_stored"]).to("cuda").transpose(1, 2)
    want = model(x).cpu()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        torch.save({"sd": {k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()}, "x": torch.from_numpy(g["t321.x_stored"])},
                   os.path.join(d, "in.pt"))
        code = (
            "import sys, torch; sys.path.insert(0, %r)\n"
            "from dfa_amd.model import CNN2D\n"
            "blob = torch.load(%r)\n"
            "m = CNN2D(in_features=180, precision='bf16'); m.load_state_dict(blob['sd']); m = m.to('cuda').eval()\n"
            "out = m(blob['x'].to('cuda').transpose(1, 2)).cpu()\n"
            "maps = open('/proc/self/maps').read()\n"
            "assert 'roctx' in maps, 'marker library not loaded'\n"
            "torch.save(out, %r)\n" % (root, os.path.join(d, "in.pt"), os.path.join(d, "out.pt")))
        env = dict(os.environ, DFA_ROCTX="1")
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        got = torch.load(os.path.join(d, "out.pt"))
    assert torch.equal(got, want)
