"""GPU parity tests for the CNN2D hot path: HIP kernels (through the C ABI) vs the CPU oracle and the committed
golden vectors generated from the reference."""
import numpy as np
import pytest
import torch

from oracle import dfa_oracle as O

pytestmark = pytest.mark.gpu

TOL_F32 = 1e-4   # north_star: logits within 1e-4 in fp32


def _model_from_sd(sd, precision="fp32"):
    from dfa_amd.model import CNN2D
    m = CNN2D(precision=precision)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    return m.to("cuda").eval()


def _ws_views(model, x, prec_bytes, B, T, F):
    """channels-last activations left in the workspace by the last forward: a1 [B,T/2,F,32], a2 [B,T/4,F,64]."""
    from dfa_amd import _lib
    ctx = _lib.Context.get(x.device)
    H1, H2 = T // 2, T // 4
    n1 = B * H1 * F * 32 * prec_bytes
    off2 = (n1 + 255) // 256 * 256
    n2 = B * H2 * F * 64 * prec_bytes
    dt = torch.float32 if prec_bytes == 4 else torch.bfloat16
    ws = ctx._ws
    a1 = ws[:n1].view(dt).view(B, H1, F, 32).float().cpu().numpy()
    a2 = ws[off2:off2 + n2].view(dt).view(B, H2, F, 64).float().cpu().numpy()
    return a1, a2


@pytest.mark.parametrize("tag", ["t321", "t64", "t7"])
def test_cnn2d_fp32_matches_golden(golden, tag):
    sd, g = golden("cnn2d_eval")
    model = _model_from_sd(sd)
    stored = torch.from_numpy(g[f"{tag}.x_stored"]).to("cuda")
    x = stored.transpose(1, 2)                                  # strided view (src/predict.py:105)
    logits, emb = model(x, return_embedding=True)
    assert logits.shape == g[f"{tag}.logits"].shape
    np.testing.assert_allclose(logits.cpu().numpy(), g[f"{tag}.logits"], atol=TOL_F32, rtol=0)
    np.testing.assert_allclose(emb.cpu().numpy(), g[f"{tag}.embedding"], atol=2e-5, rtol=1e-5)
    # contiguous input gives the same answer
    logits_c = model(x.contiguous())
    np.testing.assert_allclose(logits_c.cpu().numpy(), logits.cpu().numpy(), atol=1e-6, rtol=0)


def test_cnn2d_fp32_layers_match_golden(golden):
    sd, g = golden("cnn2d_eval")
    model = _model_from_sd(sd)
    x = torch.from_numpy(g["t16.x_stored"]).to("cuda").transpose(1, 2)
    logits = model(x)
    a1, a2 = _ws_views(model, x, 4, 1, 16, 180)
    np.testing.assert_allclose(a1.transpose(0, 3, 1, 2), g["t16.a1"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(a2.transpose(0, 3, 1, 2), g["t16.a2"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(logits.cpu().numpy(), g["t16.logits"], atol=TOL_F32, rtol=0)


@pytest.mark.parametrize("B,T", [(1, 321), (5, 33), (2, 4), (3, 130)])
def test_cnn2d_fp32_matches_oracle_random_shapes(golden, B, T):
    sd, _ = golden("cnn2d_eval")
    model = _model_from_sd(sd)
    g = torch.Generator().manual_seed(100 + B * 1000 + T)
    stored = torch.randn(B, 180, T, generator=g) * 3.2 - 0.07
    want, inter = O.cnn2d_forward(sd, stored.numpy().swapaxes(1, 2), return_intermediates=True)
    logits, emb = model(stored.to("cuda").transpose(1, 2), return_embedding=True)
    np.testing.assert_allclose(emb.cpu().numpy(), inter["embedding"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(logits.cpu().numpy(), want, atol=TOL_F32, rtol=0)


def test_cnn2d_other_in_features():
    """F is a runtime dimension (strips of 32 columns with a ragged tail): try F=40 and F=65."""
    from dfa_amd.model import CNN2D
    for F in (40, 65):
        torch.manual_seed(F)
        m = CNN2D(in_features=F).to("cuda").eval()
        with torch.no_grad():
            m.classifier.weight.mul_(30.0)
        sd = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
        x = torch.randn(2, 21, F)
        want = O.cnn2d_forward(sd, x.numpy())
        got = m(x.to("cuda")).cpu().numpy()
        np.testing.assert_allclose(got, want, atol=TOL_F32, rtol=0)


def test_cnn2d_bf16_mode_close_and_rank_preserving(golden):
    sd, g = golden("cnn2d_eval")
    model = _model_from_sd(sd, precision="bf16")
    stored = torch.from_numpy(g["t321.x_stored"]).to("cuda")
    want = g["t321.logits"]
    got_f32in = model(stored.transpose(1, 2)).cpu().numpy()
    got_bf16in = model(stored.to(torch.bfloat16).transpose(1, 2)).cpu().numpy()
    # bf16 storage / fp32 accumulate: tolerance 2e-2 relative to |logit| ~ 3 (not the 1e-4 parity mode)
    np.testing.assert_allclose(got_bf16in, want, atol=0.10, rtol=0)
    # bf16 mode stores the features in bf16 too: fp32 features are rounded (RNE) as the fused kernel loads them, which
    # must be exactly what a caller-side .to(bfloat16) gives
    assert np.array_equal(got_f32in, got_bf16in)


def test_cnn2d_fused_blocks_1_2_match_two_kernel_path_and_oracle(golden):
    """bf16 mode on bf16 features runs blocks 1+2 as one kernel (conv12_fused.hip: block 1 on the matrix cores from
    hi/lo-split weights, a1 kept in LDS).  It must agree with the two-kernel path to well inside the bf16-mode error
    (the only difference is the summation order inside block 1, which flips a few bf16 roundings of a1), for both
    feature layouts, ragged widths (F = 65, 40: last strip 5 / 10 columns), short T, and batch-independently."""
    from dfa_amd import _lib
    from dfa_amd.model import CNN2D
    sd, g = golden("cnn2d_eval")
    ctx = _lib.Context.get(torch.device("cuda"))
    gen = torch.Generator().manual_seed(5)
    sdn = {k: np.asarray(v) for k, v in sd.items()}
    cases = [("t321", torch.from_numpy(g["t321.x_stored"]), True), ("t7", torch.from_numpy(g["t7.x_stored"]), True),
             ("btf", torch.randn(3, 50, 180, generator=gen), False), ("F65", torch.randn(2, 65, 33, generator=gen), True),
             ("F40", torch.randn(2, 40, 18, generator=gen), True), ("T4", torch.randn(2, 180, 4, generator=gen), True),
             ("F31", torch.randn(1, 31, 9, generator=gen), True), ("F29", torch.randn(3, 29, 10, generator=gen), True),
             ("F61T5", torch.randn(2, 5, 61, generator=gen), False), ("F30", torch.randn(1, 30, 322, generator=gen), True),
             ("b64", torch.randn(64, 180, 321, generator=gen) * 3, True)]
    try:
        # non-dense strides: every other utterance and a frame window of a larger stored tensor
        big = torch.randn(6, 180, 400, generator=gen).to("cuda").to(torch.bfloat16)
        view = big[::2, :, 37:358].transpose(1, 2)
        model = _model_from_sd(sd, precision="bf16")
        assert torch.equal(model(view), model(view.contiguous()))
        for name, stored, is_bft in cases:
            xb = stored.to("cuda").to(torch.bfloat16)
            xb = xb.transpose(1, 2) if is_bft else xb        # [B,T,F] view of [B,F,T] storage, or contiguous [B,T,F]
            F = xb.shape[2]
            model = _model_from_sd(sd, precision="bf16") if F == 180 else CNN2D(in_features=F, precision="bf16").to("cuda").eval()
            ctx.set_option("fuse_conv1", 0)
            l0, e0 = model(xb, return_embedding=True)
            ctx.set_option("fuse_conv1", 1)
            l1, e1 = model(xb, return_embedding=True)
            scale = max(1.0, float(l0.abs().max()))
            assert float((l0 - l1).abs().max()) <= 2e-3 * scale, name
            assert float((e0 - e1).abs().max()) <= 4e-3 * max(1.0, float(e0.abs().max())), name
            if F == 180 and xb.shape[0] <= 8:           # both paths sit at the same distance from the fp32 oracle
                ref = O.cnn2d_forward(sdn, xb.float().cpu().numpy()).reshape(-1)
                d0 = np.abs(l0.cpu().numpy().reshape(-1) - ref).max()
                d1 = np.abs(l1.cpu().numpy().reshape(-1) - ref).max()
                assert d1 <= 0.06 and d1 <= 1.5 * d0 + 1e-3, (name, d0, d1)
            x32 = stored.to("cuda")
            x32 = x32.transpose(1, 2) if is_bft else x32
            lf, ef = model(x32, return_embedding=True)  # fp32 features: rounded on load == the bf16 tensor above
            assert torch.equal(lf, l1) and torch.equal(ef, e1), name
            if name == "b64":                           # batch independence of the fused path
                one, _ = model(xb[17:18], return_embedding=True)
                assert torch.equal(one, l1[17:18])
    finally:
        ctx.set_option("fuse_conv1", 1)


def test_cnn2d_block3_16x16x32_kernel_matches_32x32x16_kernel(golden):
    """bf16 block 3 runs on v_mfma_f32_16x16x32_bf16 (conv3_m16.hip); the 32x32x16 kernel computes the same fp32
    sums in a different K order, so embeddings agree to fp32 rounding of the accumulation, not to bf16 noise."""
    from dfa_amd import _lib
    sd, g = golden("cnn2d_eval")
    ctx = _lib.Context.get(torch.device("cuda"))
    model = _model_from_sd(sd, precision="bf16")
    gen = torch.Generator().manual_seed(11)
    xs = [torch.from_numpy(g[f"{t}.x_stored"]).to("cuda").transpose(1, 2) for t in ("t321", "t7")]
    xs.append((torch.randn(32, 180, 321, generator=gen) * 3.0).to(device="cuda", dtype=torch.bfloat16).transpose(1, 2))
    try:
        for x in xs:
            ctx.set_option("block3_m16", 0)
            l0, e0 = model(x, return_embedding=True)
            ctx.set_option("block3_m16", 1)
            l1, e1 = model(x, return_embedding=True)
            assert float((e0 - e1).abs().max()) <= 1e-5 * max(1.0, float(e0.abs().max())), tuple(x.shape)
            assert float((l0 - l1).abs().max()) <= 2e-5 * max(1.0, float(l0.abs().max())), tuple(x.shape)
    finally:
        ctx.set_option("block3_m16", 1)


def test_extract_embeddings_matches_per_batch_forward_and_oracle(golden):
    """predict.extract_embeddings: the [N, 128*F] block-3 embedding export equals the per-call `return_embedding`
    output, for any batch size, and the oracle's mean-over-T feature map (src/model.py:37-38)."""
    from dfa_amd.predict import extract_embeddings, predict_scores
    sd, g = golden("cnn2d_eval")
    model = _model_from_sd(sd)
    gen = torch.Generator().manual_seed(21)
    stored = torch.randn(7, 180, 64, generator=gen)
    emb, logits = extract_embeddings(model, stored, batch_size=3)
    assert tuple(emb.shape) == (7, 128 * 180) and tuple(logits.shape) == (7,)
    lg, e = model(stored.to("cuda").transpose(1, 2), return_embedding=True)
    assert torch.equal(emb, e.cpu()) and torch.equal(logits, lg.squeeze(-1).cpu())
    assert torch.equal(logits, predict_scores(model, stored, batch_size=4, apply_sigmoid=False).cpu())
    _, inter = O.cnn2d_forward({k: np.asarray(v) for k, v in sd.items()}, stored.numpy().swapaxes(1, 2),
                               return_intermediates=True)
    np.testing.assert_allclose(emb.numpy(), inter["embedding"], atol=2e-5, rtol=1e-4)


def test_cnn2d_batch_independence_full_size(golden):
    """BASELINE configs[1] shape [256,321,180]: every utterance's logit must equal the logit it gets in a batch of
    its own (eval mode has no cross-sample op) -- a size-independent property checked at the full benchmark size."""
    sd, _ = golden("cnn2d_eval")
    model = _model_from_sd(sd)
    g = torch.Generator().manual_seed(7)
    stored = (torch.randn(256, 180, 321, generator=g) * 3.2 - 0.07).to("cuda")
    x = stored.transpose(1, 2)
    full = model(x).cpu().numpy()
    for idx in (0, 1, 100, 255):
        one = model(x[idx:idx + 1]).cpu().numpy()
        np.testing.assert_allclose(one, full[idx:idx + 1], atol=1e-6, rtol=0)
    # and a few against the oracle
    want = O.cnn2d_forward(sd, stored[:3].cpu().numpy().swapaxes(1, 2))
    np.testing.assert_allclose(full[:3], want, atol=TOL_F32, rtol=0)


def test_cnn2d_error_behaviour(golden):
    sd, _ = golden("cnn2d_eval")
    model = _model_from_sd(sd)
    with pytest.raises(ValueError):
        model(torch.zeros(2, 321, 100, device="cuda"))          # F != in_features
    with pytest.raises(ValueError):
        model(torch.zeros(2, 3, 180, device="cuda"))            # T too short for two pools
    with pytest.raises(ValueError):
        model(torch.zeros(321, 180, device="cuda"))             # not (B,T,F)
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 321, 180))                          # CPU input: no CPU path
    with pytest.raises(ValueError):
        model(torch.zeros(1, 321, 180, device="cuda", dtype=torch.float16))


def test_predict_end_to_end_eer_parity(tmp_path, golden):
    """features.pkl -> HIP predict -> prediction.pkl -> scorer, against the oracle run on the same features:
    logits within 1e-4 and IDENTICAL EER/threshold index for both sigmoid and raw-logit scores (configs[0])."""
    import pandas as pd
    from dfa_amd import evaluation, predict
    from oracle import torch_ref as R
    sd, _ = golden("cnn2d_eval")
    n = 96
    g = torch.Generator().manual_seed(5)
    labels = (torch.rand(n, generator=g) > 0.55).long()
    # class-1 utterances carry a low-rank pattern so that scores separate only partly (EER strictly inside (0, 1))
    pattern = torch.outer(torch.sin(torch.arange(180) / 7.0), torch.cos(torch.arange(321) / 23.0))
    feats = [torch.randn(180, 321, generator=g) * 3.2 - 0.07 + 0.35 * labels[i] * pattern for i in range(n)]
    fdf = pd.DataFrame({"uttid": [f"u{i:05d}" for i in range(n)], "features": feats})
    ldf = pd.DataFrame({"uttid": [f"u{i:05d}" for i in range(n)], "label": labels.numpy()})
    fp, lp, ck = str(tmp_path / "features.pkl"), str(tmp_path / "labels.pkl"), str(tmp_path / "cnn2d.pt")
    fdf.to_pickle(fp)
    ldf.to_pickle(lp)
    torch.save({"model_state": {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}}, ck)
    ref_logits = R.cnn2d_forward(sd, torch.stack(feats).transpose(1, 2)).squeeze(-1).double()
    for flag, ref_scores in (([], torch.sigmoid(ref_logits.float()).double()), (["--no-apply-sigmoid"], ref_logits)):
        out = str(tmp_path / f"prediction{len(flag)}.pkl")
        predict.main(["--features", fp, "--checkpoint", ck, "--model", "cnn2d", "--out", out, "--batch-size", "32"]
                     + flag)
        got = pd.read_pickle(out)
        assert list(got["uttid"]) == list(fdf["uttid"])
        np.testing.assert_allclose(got["predictions"].values, ref_scores.numpy(), atol=TOL_F32, rtol=0)
        res = evaluation.score_prediction_file(out, lp)
        want = O.calculate_eer(ref_scores.tolist(), labels.tolist())
        assert res["eer"] == want[0], (res, want)
        assert 0.0 < res["eer"] < 1.0          # non-degenerate: the metric is sensitive to rank changes
    # the bf16 storage mode (the headline configuration): scores within the bf16-mode tolerance of the reference's, and
    # the EER moves by at most one rank swap of the 96 utterances
    out16 = str(tmp_path / "prediction_bf16.pkl")
    predict.main(["--features", fp, "--checkpoint", ck, "--model", "cnn2d", "--out", out16, "--batch-size", "32",
                  "--no-apply-sigmoid", "--precision", "bf16"])
    got16 = pd.read_pickle(out16)
    np.testing.assert_allclose(got16["predictions"].values, ref_logits.numpy(), atol=0.1, rtol=0)
    # --embeddings-out: the [N, 128*F] export next to the predictions, consistent with the scores it came with
    eo = str(tmp_path / "emb.pt")
    predict.main(["--features", fp, "--checkpoint", ck, "--model", "cnn2d", "--out", str(tmp_path / "p3.pkl"), "--batch-size",
                  "32", "--no-apply-sigmoid", "--embeddings-out", eo])
    blob = torch.load(eo)
    assert list(blob["uttid"]) == list(fdf["uttid"]) and tuple(blob["embeddings"].shape) == (n, 128 * 180)
    np.testing.assert_allclose(blob["logits"].numpy(), ref_logits.numpy(), atol=TOL_F32, rtol=0)
    eer16 = evaluation.score_prediction_file(out16, lp)["eer"]
    eer32 = O.calculate_eer(ref_logits.tolist(), labels.tolist())[0]
    assert abs(eer16 - eer32) <= 1.0 / min(int(labels.sum()), n - int(labels.sum())) + 1e-12, (eer16, eer32)


def test_cnn2d_lds_dma_staging_matches_register_staging(golden):
    """The LDS-DMA (global_load_lds) input staging must give bit-identical results to register staging."""
    from dfa_amd import _lib
    sd, g = golden("cnn2d_eval")
    ctx = _lib.Context.get(torch.device("cuda"))
    try:
        ctx.set_option("fuse_conv1", 0)      # the stand-alone 32x32x16 kernels are the ones with both staging paths
        ctx.set_option("block3_m16", 0)
        for prec in ("fp32", "bf16"):
            model = _model_from_sd(sd, precision=prec)
            for tag in ("t321", "t7"):
                x = torch.from_numpy(g[f"{tag}.x_stored"]).to("cuda").transpose(1, 2)
                ctx.set_option("conv_dma", 0)
                ref_l, ref_e = model(x, return_embedding=True)
                ctx.set_option("conv_dma", 1)
                dma_l, dma_e = model(x, return_embedding=True)
                assert torch.equal(ref_e, dma_e) and torch.equal(ref_l, dma_l), (prec, tag)
    finally:
        ctx.set_option("conv_dma", -1)
        ctx.set_option("fuse_conv1", 1)
        ctx.set_option("block3_m16", 1)


def test_cnn2d_product_kernels_match_their_compiler_scheduled_twins(golden):
    """The two kernels of the headline path -- conv12_fused_kernel and conv3_m16_meant_kernel -- read their LDS operands
    through the asm pipeline; with lds_pipe = 0 the same kernels run with compiler-scheduled LDS loads.  Same MFMAs in
    the same order: logits and embeddings must be bit-identical, for bf16 and fp32 features, ragged widths included."""
    from dfa_amd import _lib
    from dfa_amd.model import CNN2D
    sd, g = golden("cnn2d_eval")
    ctx = _lib.Context.get(torch.device("cuda"))
    gen = torch.Generator().manual_seed(13)
    cases = [torch.from_numpy(g["t321.x_stored"]).to("cuda").to(torch.bfloat16).transpose(1, 2),
             torch.from_numpy(g["t7.x_stored"]).to("cuda").transpose(1, 2),
             (torch.randn(200, 180, 321, generator=gen) * 3.0).to("cuda", dtype=torch.bfloat16).transpose(1, 2),
             torch.randn(3, 65, 40, generator=gen).to("cuda", dtype=torch.bfloat16).transpose(1, 2)]
    try:
        for x in cases:
            F = x.shape[2]
            model = _model_from_sd(sd, precision="bf16") if F == 180 else CNN2D(in_features=F, precision="bf16").to("cuda").eval()
            ctx.set_option("lds_pipe", 0)
            l0, e0 = model(x, return_embedding=True)
            ctx.set_option("lds_pipe", 1)
            l1, e1 = model(x, return_embedding=True)
            assert torch.equal(l0, l1) and torch.equal(e0, e1), tuple(x.shape)
    finally:
        ctx.set_option("lds_pipe", 1)


def test_cnn2d_pipelined_lds_reads_match_compiler_scheduled_twins(golden):
    """The asm-pipelined fragment reads (conv3x3_mfma.h, PFD > 0) change the instruction schedule only: every pipelined
    bf16 kernel must be bit-identical to its compiler-scheduled twin (lds_pipe = 0), for both staging paths, at small
    sizes and at the headline batch."""
    from dfa_amd import _lib
    sd, g = golden("cnn2d_eval")
    ctx = _lib.Context.get(torch.device("cuda"))
    model = _model_from_sd(sd, precision="bf16")
    gen = torch.Generator().manual_seed(7)
    big = (torch.randn(256, 180, 321, generator=gen) * 3.0).to(device="cuda", dtype=torch.bfloat16).transpose(1, 2)
    xs = [torch.from_numpy(g[f"{t}.x_stored"]).to("cuda").transpose(1, 2) for t in ("t321", "t7")] + [big]
    try:
        ctx.set_option("fuse_conv1", 0)          # the stand-alone block-2 kernel has the twins (bf16 inputs would fuse it away)
        ctx.set_option("block3_m16", 0)          # ... and so does the 32x32x16 block-3 kernel
        for dma in (0, 1):
            ctx.set_option("conv_dma", dma)
            for x in xs:
                ctx.set_option("lds_pipe", 0)
                ref_l, ref_e = model(x, return_embedding=True)
                ctx.set_option("lds_pipe", 1)
                got_l, got_e = model(x, return_embedding=True)
                assert torch.equal(ref_e, got_e) and torch.equal(ref_l, got_l), (dma, tuple(x.shape))
    finally:
        ctx.set_option("conv_dma", -1)
        ctx.set_option("lds_pipe", 1)
        ctx.set_option("fuse_conv1", 1)
        ctx.set_option("block3_m16", 1)
