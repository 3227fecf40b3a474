"""CPU tests of the host side: drop-in module surface, scorer, fusion, datasets/loaders, prediction files,
checkpoints, and that libdfa_hip.so loads and exports every symbol include/dfa_hip.h declares (no compute calls)."""
import os
import re
import types

import numpy as np
import pandas as pd
import pytest
import torch

import dfa_amd  # noqa: F401
from dfa_amd import _lib, evaluation, fusion
from dfa_amd.dataset import AudioDeepfakeDataset
from dfa_amd.dataloaders import FlatBatcher, create_dataloaders, make_loader
from dfa_amd.model import CNN2D
from dfa_amd.predict import write_predictions
from dfa_amd.training import load_checkpoint, save_checkpoint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------ C ABI surface
def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "dfa_hip.h")).read()
    declared = set(re.findall(r"\b(dfa_[a-z0-9_]+)\s*\(", header))
    declared.discard("dfa_ctx")
    assert declared, "no prototypes found in include/dfa_hip.h"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), f"libdfa_hip.so does not export {name}"
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert declared == bound, f"ctypes table and header disagree: {declared ^ bound}"
    assert lib.dfa_version() == 100
    assert lib.dfa_error_name(-1) == b"DFA_E_BAD_SHAPE"
    assert lib.dfa_workspace_bytes(None, _lib.MODEL_CNN2D, 256, 321, 180, _lib.PREC_BF16) > 0


# ------------------------------------------------------------------------------------------------ module surface
def test_cnn2d_state_dict_matches_reference_layout(golden):
    sd, _ = golden("cnn2d_eval")
    m = CNN2D()
    mine = m.state_dict()
    assert list(mine.keys()) == list(sd.keys())              # same keys in the same order as the reference
    for k, v in sd.items():
        assert tuple(mine[k].shape) == tuple(v.shape), k
    assert mine["conv.1.num_batches_tracked"].dtype == torch.int64
    assert sum(p.numel() for p in m.parameters()) == 116_161
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    assert [n for n, _ in m.named_parameters()][:4] == ["conv.0.weight", "conv.0.bias", "conv.1.weight", "conv.1.bias"]


def test_cnn2d_default_init_is_bit_identical_to_reference(golden):
    """make_golden.py built the reference CNN2D under torch.manual_seed(7): same seed here -> same conv weights."""
    _, g = golden("cnn2d_train")
    torch.manual_seed(7)
    m = CNN2D(in_features=180, dropout=0.0)
    sd = m.state_dict()
    for k in ("conv.0.weight", "conv.0.bias", "conv.5.weight", "conv.5.bias", "conv.10.weight", "conv.10.bias",
              "classifier.bias"):
        np.testing.assert_array_equal(sd[k].numpy(), g["init.sd." + k])
    np.testing.assert_allclose(sd["classifier.weight"].numpy() * 40.0, g["init.sd.classifier.weight"], rtol=1e-6)


def test_no_cpu_fallback():
    m = CNN2D().eval()
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 321, 180))
    with pytest.raises(ValueError):
        m(torch.zeros(321, 180))
    with pytest.raises(ValueError):
        CNN2D(num_classes=3)
    with pytest.raises(ValueError):
        CNN2D(precision="fp8")


# ------------------------------------------------------------------------------------------------ scorer / fusion
@pytest.mark.parametrize("k", ["sep", "mix", "inv", "one", "tie", "rng"])
def test_calculate_eer_matches_reference(golden, k):
    _, g = golden("host")
    s, l = g[f"eer.{k}.scores"], g[f"eer.{k}.labels"]
    assert evaluation.calculate_eer(s.tolist(), l.tolist()) == tuple(g[f"eer.{k}.result"])
    thr = g[f"eer.{k}.result"][1]
    assert tuple(float(v) for v in evaluation.confusion_at_threshold(s, l, thr)) == tuple(g[f"eer.{k}.confusion"])


def test_fusion_matches_reference(golden):
    _, g = golden("host")
    np.testing.assert_array_equal(fusion.normalise_scores(g["fuse.sup"]), g["fuse.sup_norm"])
    np.testing.assert_array_equal(fusion.normalise_scores(np.full(5, 0.25)), g["fuse.const_norm"])
    table, best_eer, best_alpha = fusion.alpha_sweep(g["fuse.sup"], g["fuse.cae"], g["fuse.labels"].tolist())
    np.testing.assert_array_equal(np.array(table), g["fuse.table"])
    assert best_eer == g["fuse.table"][:, 1].min()
    np.testing.assert_array_equal(fusion.ensemble_mean([g["fuse.sup"], g["fuse.cae"]]), g["fuse.ens_mean"])
    np.testing.assert_array_equal(fusion.hybrid_scores(g["fuse.sup"], g["fuse.cae"], 1.0), g["fuse.sup_norm"])
    np.testing.assert_array_equal(fusion.gather_sharded([1.0, 2.0]), np.array([1.0, 2.0]))


# ------------------------------------------------------------------------------------------------ data
def _write_pickles(tmp_path, n=10, T=321, with_uttid=True):
    g = torch.Generator().manual_seed(0)
    rows = [{"uttid": f"utt{i:04d}", "features": torch.randn(180, T, generator=g)} for i in range(n)]
    feats = pd.DataFrame(rows)
    if not with_uttid:
        feats = feats.drop(columns=["uttid"])
    labels = pd.DataFrame({"uttid": [f"utt{i:04d}" for i in reversed(range(n))], "label": [i % 2 for i in range(n)]})
    fp, lp = str(tmp_path / "features.pkl"), str(tmp_path / "labels.pkl")
    feats.to_pickle(fp)
    labels.to_pickle(lp)
    return fp, lp, feats, labels


def test_dataset_and_loaders(tmp_path):
    fp, lp, feats, labels = _write_pickles(tmp_path)
    ds = AudioDeepfakeDataset(fp, lp)
    assert len(ds) == 10
    f0, l0 = ds[0]
    assert f0.shape == (180, 321) and f0.dtype == torch.float32 and l0.dtype == torch.float32
    want = labels.set_index("uttid").loc["utt0000", "label"]
    assert float(l0) == float(want)                                   # merged on uttid, not on position
    stack, lab = ds.stacked()
    assert stack.shape == (10, 180, 321) and lab.shape == (10,)
    loader = make_loader(fp, lp, batch_size=4, num_workers=0)
    fb, lb = next(iter(loader))
    assert fb.shape == (4, 180, 321) and lb.shape == (4,)
    tr, dv, te = create_dataloaders(fp, lp, fp, lp, fp, batch_size=4, num_workers=0)
    assert next(iter(te)).shape == (4, 180, 321)                      # test split: features only
    batches = list(FlatBatcher(stack, lab, 4, device="cpu"))
    assert [b[0].shape[0] for b in batches] == [4, 4, 2]
    assert torch.equal(torch.cat([b[0] for b in batches]), stack)
    # contiguous utterance shards for data-parallel inference
    parts = [torch.cat([b[0] for b in FlatBatcher(stack, lab, 4, device="cpu", rank=r, world=3)]) for r in range(3)]
    assert [p.shape[0] for p in parts] == [4, 4, 2] and torch.equal(torch.cat(parts), stack)
    evaluation.verify_uttid_alignment(fp, lp)


def test_dataset_errors(tmp_path):
    fp, lp, _, _ = _write_pickles(tmp_path, with_uttid=False)
    with pytest.raises(ValueError):
        AudioDeepfakeDataset(fp, lp)
    with pytest.raises(ValueError):
        evaluation.verify_uttid_alignment(fp, lp)


def test_prediction_file_schema_and_scoring(tmp_path):
    fp, lp, feats, labels = _write_pickles(tmp_path)
    out = str(tmp_path / "prediction.pkl")
    scores = np.linspace(0.0, 1.0, 10)
    df = write_predictions(feats["uttid"].values, scores, out)
    back = pd.read_pickle(out)
    assert list(back.columns) == ["uttid", "predictions"] and back["predictions"].dtype == np.float64
    assert back.equals(df)
    res = evaluation.score_prediction_file(out, lp)
    merged = pd.merge(back, labels, on="uttid")
    assert (res["eer"], res["threshold"]) == evaluation.calculate_eer(merged["predictions"].values,
                                                                      merged["label"].values)
    with pytest.raises(ValueError):
        write_predictions(feats["uttid"].values, scores[:-1], out)
    labels.iloc[:-1].to_pickle(lp)
    with pytest.raises(ValueError):
        evaluation.score_prediction_file(out, lp)


def test_checkpoint_roundtrip(tmp_path):
    m = CNN2D()
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min")
    args = types.SimpleNamespace(model="cnn2d", batch_size=32, lr=1e-3, dropout=0.2, in_features=180)
    path = str(tmp_path / "run" / "cnn2d_best.pt")
    save_checkpoint(m, opt, 3, args, path, scheduler=sched)
    m2 = CNN2D()
    blob = load_checkpoint(path, model=m2)
    assert set(blob) == {"model_state", "optimizer_state", "epoch", "config", "scheduler_state"}
    assert blob["epoch"] == 3 and blob["config"]["model_name"] == "cnn2d" and blob["config"]["hidden_dim"] is None
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    with pytest.raises(FileNotFoundError):
        load_checkpoint(str(tmp_path / "missing.pt"))


def test_flat_ingest_roundtrip(tmp_path):
    from dfa_amd import ingest
    fp, lp, feats, labels = _write_pickles(tmp_path, n=6)
    meta = ingest.convert(fp, str(tmp_path / "flat"), lp, dtype="fp32")
    ff = ingest.FlatFeatures(str(tmp_path / "flat"))
    ds = AudioDeepfakeDataset(fp, lp)
    stack, lab = ds.stacked()
    assert len(ff) == 6 and ff.uttids == list(ds.uttids()) and meta["shape"] == [6, 180, 321]
    assert torch.equal(ff.tensor(), stack) and torch.equal(ff.labels, lab)
    ingest.convert(fp, str(tmp_path / "flat16"), None, dtype="bf16")
    f16 = ingest.FlatFeatures(str(tmp_path / "flat16"))
    assert f16.tensor().dtype == torch.bfloat16 and f16.labels is None
    assert torch.equal(f16.tensor(), torch.stack(list(feats["features"])).to(torch.bfloat16))
    parts = list(FlatBatcher(ff.tensor(), ff.labels, 4, device="cpu"))
    assert [b[0].shape[0] for b in parts] == [4, 2]


def test_pipelined_lds_reads_are_never_touched_in_flight():
    """Static check of the compiled gfx950 assembly (tools/check_lds_pipeline.py): between an asm-issued ds_read_b128
    and the counted s_waitcnt that retires it, no instruction may read or write the destination registers."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_lds_pipeline", os.path.join(root, "tools", "check_lds_pipeline.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    if not os.path.exists(chk.HIPCC):
        pytest.skip("hipcc not available")
    csrc = os.path.join(root, "deep-fake-audio-classifier_amd", "csrc")
    total = 0
    known = "conv3x3_mfma_kernelIfLi64ELi4ELi1ELi1ELi1ELi1ELi1ELb1ELb0ELb0ELi3E"   # the diagnostic instantiation (variant 7)
    flagged_known, nmfma = 0, 0
    for name in ("conv3x3_inst_cnn2d.hip", "conv3x3_inst_cae.hip", "conv12_fused.hip", "conv3_m16.hip", "wgrad_mfma.hip",
                 "conv3x3_inst_train.hip", "conv_split.hip"):
        asm = chk.compile_to_asm(os.path.join(csrc, name))
        kernels, nreads, violations = chk.check_asm(asm)
        assert not violations, violations[:5]
        total += nreads
        # second rule: a weight fragment half-reassembled across loop phases (the hipcc miscompile behind round 1's "wrong
        # sums" of the pipelined fp32 ACCIN kernel).  It must flag the diagnostic instantiation kept for that purpose -- the
        # positive control, confirmed wrong on the GPU by tools/gpu_accin_probe.py -- and nothing that ships.
        nk, nm, v2 = chk.check_operand_provenance(asm)
        nmfma += nm
        flagged_known += sum(1 for k, _, _ in v2 if known in k)
        others = [v for v in v2 if known not in v[0]]
        assert not others, others[:5]
    assert total > 0          # the pipelined instantiations exist
    assert nmfma > 10000 and flagged_known >= 1


def test_checker_flags_a_touched_in_flight_buffer_load():
    """Positive and negative control of the checker's vector-memory rule on hand-written assembly: the destination of an asm
    buffer load may not be read before the s_waitcnt vmcnt(N) that retires it (loads return in order), nor be in flight over a
    loop back-edge."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_lds_pipeline", os.path.join(root, "tools", "check_lds_pipeline.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)

    def kernel(body):
        return "_Z1kv:\n" + "\n".join("\t" + l for l in body) + "\n\ts_endpgm\n"

    load = lambda lo, hi: [";;#ASMSTART", "buffer_load_dwordx4 v[%d:%d], v1, s[4:7], 0 offen" % (lo, hi), ";;#ASMEND"]
    wait = lambda n: [";;#ASMSTART", "s_waitcnt vmcnt(%d)" % n, ";;#ASMEND"]
    good = load(10, 13) + load(14, 17) + ["v_add_u32_e32 v2, v3, v4"] + wait(1) + ["ds_write_b128 v5, v[10:13]"] + wait(0) + \
        ["ds_write_b128 v5, v[14:17]"]
    assert chk.check_asm(kernel(good))[2] == []
    early = load(10, 13) + load(14, 17) + wait(1) + ["ds_write_b128 v5, v[14:17]"]        # the second load is still in flight
    assert any("in-flight buffer-load" in v[2] for v in chk.check_asm(kernel(early))[2])
    copied = load(10, 13) + ["v_mov_b32_e32 v20, v11"] + wait(0)                            # the compiler copying a stale register
    assert any("in-flight buffer-load" in v[2] for v in chk.check_asm(kernel(copied))[2])
    looped = [".LBB0_1:"] + load(10, 13) + ["s_cbranch_vccnz .LBB0_1"] + wait(0)
    assert any("back-edge" in v[2] for v in chk.check_asm(kernel(looped))[2])


def test_m16_swizzle_is_conflict_free():
    """conv3_m16.hip stores chunk c of pixel slot s at physical chunk c ^ (s & 6).  gfx950 services a ds_read_b128 in
    four fixed groups of 16 lanes; within a group the 16 x 16-byte accesses must fall in 16 distinct bank quads
    ((address / 16) mod 16).  Exhaustive over tap column, pixel tile, k-group and lane group."""
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
              list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
              list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
    for dx in range(3):
        for pb in range(2):
            for kk in range(2):
                for grp in groups:
                    quads = set()
                    for lane in grp:
                        p, q = lane & 15, lane >> 4
                        slot = 16 * pb + p + dx
                        addr = slot * 128 + (((4 * kk + q) ^ (slot & 6)) << 4)
                        quads.add((addr // 16) % 16)
                    assert len(quads) == 16, (dx, pb, kk)


def test_split_kernel_swizzles_are_conflict_free():
    """conv_split.hip keeps split pixels ([hi C bf16][lo C bf16]) in LDS with chunk c at physical chunk c ^ swz(slot):
    swz = slot & 6 for 128-byte pixels (C = 32), (slot & 7) << 1 for 256-byte pixels (C = 64).  Within each of the four
    16-lane groups gfx950 services a ds_read_b128 in, the 16 accesses must fall in 16 distinct bank quads ((addr / 16) mod
    16).  Exhaustive over tap column, pixel tile, k-step, hi/lo half and lane group."""
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
              list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
              list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
    for cin, swz in ((32, lambda s: s & 6), (64, lambda s: (s & 7) << 1)):
        pb_bytes, cpp, kkn = cin * 4, cin // 4, cin // 32
        for dx in range(3):
            for tile in range(2):
                for kk in range(kkn):
                    for hl in range(2):
                        c0 = hl * (cpp // 2) + 4 * kk
                        for grp in groups:
                            quads = set()
                            for lane in grp:
                                p, q = lane & 15, lane >> 4
                                slot = 16 * tile + p + dx
                                # the kernel's address: (slot*PB + ((q ^ swz(p+dx)) << 4)) ^ (c0 << 4), tile = +16*PB
                                addr = ((p + dx) * pb_bytes + ((q ^ swz(p + dx)) << 4)) ^ (c0 << 4)
                                addr += 16 * tile * pb_bytes
                                assert addr // pb_bytes == slot                      # stays inside its pixel slot
                                assert (addr % pb_bytes) // 16 == (c0 + q) ^ swz(slot)  # = logical chunk ^ swizzle
                                quads.add((addr // 16) % 16)
                            assert len(quads) == 16, (cin, dx, tile, kk, hl)


def _run_asan_harness():
    """build (host-only, -fsanitize=address) and run tests/host/asan_harness.cpp; returns (returncode, output)"""
    import subprocess
    csrc = os.path.join(ROOT, "deep-fake-audio-classifier_amd", "csrc")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    subprocess.run(["make", "-C", csrc, "asan", "-j", "8"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    rt = subprocess.run([hipcc, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.dirname(rt) + ":" + env.get("LD_LIBRARY_PATH", "")
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0"       # (the HIP runtime's own start-up allocations are not ours to judge)
    r = subprocess.run([os.path.join(ROOT, "deep-fake-audio-classifier_amd", "lib", "dfa_asan_harness")], env=env,
                       capture_output=True, text=True, timeout=600)
    return r.returncode, r.stdout + r.stderr


def test_host_layer_under_address_sanitizer():
    """SURVEY.md section 5 (sanitizers): the host side of the C ABI -- every workspace planner over a sweep of shapes, the error
    tables, option parsing and each entry point's null-context path -- runs clean under AddressSanitizer in a host-only build of
    csrc/*.hip (no kernels; `make asan`).  GPU AddressSanitizer is not available on the pool; the device-present half of the
    harness (argument checks in front of the launches) runs in the GPU suite."""
    rc, out = _run_asan_harness()
    assert rc == 0 and "AddressSanitizer" not in out, out[-3000:]
    assert "0 failure(s)" in out


@pytest.mark.gpu
def test_host_layer_under_address_sanitizer_with_a_device():
    rc, out = _run_asan_harness()
    assert rc == 0 and "AddressSanitizer" not in out, out[-3000:]
    assert "device present, 0 failure(s)" in out


def test_resident_batcher_yields_the_same_batches_as_the_host_fed_loader():
    """Round 3: `dataloaders.ResidentBatcher` (the training set uploaded once, batches = device-side row gathers; SURVEY section 8(e)'s
    loader, MI355X-first: 288 GB of HBM hold the reference's sets many times over) against `IndexedFlatBatcher` on the same index
    order -- same rows, same labels, ragged last batch included; optional storage dtype (bf16: the rounding the first kernel would
    apply on load).  CPU tensors here (the classes are device-agnostic); the two-rank CLI tests run it on the GPU."""
    import torch
    from dfa_amd.dataloaders import IndexedFlatBatcher, ResidentBatcher
    g = torch.Generator().manual_seed(4)
    feats = torch.randn(37, 6, 5, generator=g)
    labels = (torch.rand(37, generator=g) > 0.5).float()
    perm = torch.randperm(37, generator=g)
    res = ResidentBatcher(feats, labels, 8, device="cpu", chunk_rows=10)
    got = list(res.epoch(perm))
    want = list(IndexedFlatBatcher(feats, labels, perm, 8, device="cpu"))
    assert len(got) == len(want) == len(res.epoch(perm)) == 5
    for (f0, l0), (f1, l1) in zip(got, want):
        assert torch.equal(f0, f1) and torch.equal(l0, l1)
    assert got[-1][0].shape[0] == 5                                        # 37 = 4 x 8 + 5
    res16 = ResidentBatcher(feats, None, 8, device="cpu", dtype=torch.bfloat16)
    f16, l16 = next(iter(res16.epoch(perm)))
    assert l16 is None and f16.dtype == torch.bfloat16 and torch.equal(f16, feats[perm[:8]].to(torch.bfloat16))
    assert res16.bytes_resident == 37 * 30 * 2 and not ResidentBatcher.fits(feats, "cpu")
