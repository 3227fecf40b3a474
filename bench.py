#!/usr/bin/env python3
"""Headline benchmark: utterances/s of the CNN2D eval forward (BASELINE.json configs[1]: synthetic [256,321,180],
bf16 storage / fp32 accumulate) on N MI355X GPUs of one node, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of 256 utterances per GPU, inputs already resident in HBM.
Inference shards by utterance with no data-path collective, so N GPUs run N independent
shards ("scaling": "weak"); the only collectives are the timing barrier and the max-over-ranks of the elapsed time.
`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches the torch.distributed.run command above
itself, as a child process, before anything in this process touches the GPU, and exits with the child's code.
Rank 0 prints ONE JSON line.  Extra objects on it:
  roofline     -- the dominant kernel (block-3 MFMA conv): algorithmic FLOPs per launch / average launch duration
                  measured with HIP events on the launch stream during the timed steps, vs the dense MFMA peak
  cpu_baseline -- the CPU restatement of the reference path (oracle/torch_ref.py) timed on this host, rank 0, N=1
  parity_fast  -- the same workload in the split-bf16 parity mode (DFA_PREC_BF16X3: logits within 1e-4 of the reference)
  fp32_parity  -- the same workload in the exact-fp32 parity mode (logits within 1e-4 of the reference)
  train_step   -- BASELINE configs[2] on one GPU with its own roofline object (9,621,849,600 FLOP per utterance)
  end_to_end   -- features.pkl on the host -> prediction.pkl (unpickle, H2D over PCIe, kernels, D2H, pickle) on the GPU
                  path and (cpu_baseline.end_to_end) on the CPU restatement of src/predict.py, same file; plus the
                  flat-file ingest rate (memmap -> pinned double-buffered H2D -> kernels)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "utterances/sec fwd (2D-CNN, [B,T=321,F=180]) at 1/2/4/8 GPU; dev EER parity"
B_PER_GPU, T, F = 256, 321, 180
FLOPS_PER_UTT = 3_218_376_960          # 2 x MAC, conv + linear
BLOCK3_KERNEL = {"bf16": "conv3_m16_meant_kernel", "fp32": "conv3x3_mfma_kernel<float,64,...>",
                 "bf16x3": "conv_split_kernel<64,8,MEAN_T>"}
BLOCK3_FLOPS_PER_UTT = 2 * 1_061_683_200  # Conv2d 64->128 on (80,180): the dominant kernel (66 % of the FLOPs)
# dense MFMA peaks, MI355X_MICROARCH.md "Chip-level parameters".  bf16x3 issues THREE bf16 MFMAs per algorithmic product
# (hi*hi + lo*hi + hi*lo), so its ceiling in algorithmic FLOP/s is a third of the bf16 peak.
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3, "bf16x3": 2500.0 / 3.0}


PMC_FILES = ("r03_pmc_traffic.json", "r02_pmc_traffic.json")    # newest first; the file used is named in the JSON line
PMC_TAGS = {"bf16": "conv3_m16_meant_kernel<true, false>", "fp32": "conv3x3_mfma_kernel<float, 64, 4",
            "bf16x3": "conv_split_kernel<64, 8, 1"}


def _pmc_blob():
    for name in PMC_FILES:
        try:
            return name, json.load(open(os.path.join(ROOT, "profiles", name)))
        except (OSError, ValueError):
            continue
    return None, {}


def pmc_kernel_traffic(tag):
    """(HBM bytes per launch, source file) of the kernel whose name contains `tag`, from the committed rocprofv3 PMC passes
    (profiles/rNN_pmc_traffic.json, made by tools/pmc_traffic_json.py from separate --pmc runs: FETCH_SIZE x 2 + WRITE_SIZE per
    MI355X_MICROARCH.md).  Hardware counters cannot be read from inside this process, so this is evidence about the PROFILED
    build: the source file is named next to the number, and a kernel the newest summary does not hold gives (None, file) --
    never a silent fall-back to an older round's file."""
    name, blob = _pmc_blob()
    for kname, rec in blob.get("kernels", {}).items():
        if tag in kname and rec.get("hbm_bytes_per_launch"):
            return rec["hbm_bytes_per_launch"], name
    if tag == PMC_TAGS["fp32"]:           # the fp32 block-3 kernel is unchanged since round 1 and was profiled there only
        try:
            blob = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            for kname, rec in blob.get("kernels", {}).items():
                if tag in kname and rec.get("hbm_bytes_per_launch"):
                    return rec["hbm_bytes_per_launch"], "r01_pmc_traffic.json"
        except (OSError, ValueError):
            pass
    return None, name


def pmc_step_traffic(step):
    """HBM bytes of one whole step ("train_step", "cnn1d_fwd", "cae_score", ...) summed over its kernels, from the same file."""
    name, blob = _pmc_blob()
    rec = blob.get("steps", {}).get(step)
    return (rec.get("hbm_bytes_per_step") if rec else None), name


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    # defaults: 200 timed steps after 50 warm-up steps (0.2 s of GPU time).  The first ~20 launches after an idle period
    # run ~9 % slower (clock ramp): 20/5 reads 344 k utt/s where 100/20 and 400/100 read 374-377 k on the same box.
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=50)
    p.add_argument("--batch", type=int, default=B_PER_GPU, help="utterances per GPU per step")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-train", action="store_true", help="skip the training-step leg")
    p.add_argument("--no-other-models", action="store_true", help="skip the CNN1D / auto-encoder legs")
    p.add_argument("--small-batch", action="store_true", help="add the batch 1 / batch 32 latency leg (time-axis split on / off)")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the CPU baseline sample")
    p.add_argument("--e2e-utts", type=int, default=512, help="utterances in the end-to-end (features.pkl -> prediction.pkl) legs")
    return p.parse_args()


def build_model(torch, device, precision):
    from dfa_amd.model import CNN2D
    torch.manual_seed(0)
    model = CNN2D(in_features=F, precision=precision)
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():                      # non-trivial BN statistics instead of the (0, 1) of a fresh init
        for i in (1, 6, 11):
            bn = model.conv[i]
            bn.running_mean.copy_(0.2 * torch.randn(bn.running_mean.shape, generator=g))
            bn.running_var.copy_(0.5 + torch.rand(bn.running_var.shape, generator=g))
            bn.weight.copy_(0.5 + torch.rand(bn.weight.shape, generator=g))
            bn.bias.copy_(0.1 * torch.randn(bn.bias.shape, generator=g))
        model.classifier.weight.mul_(50.0)
    return model.to(device).eval()


def precondition(torch, ctx, model, x, slot=2, batch=25, tol=0.01, max_batches=40):
    """Untimed launches until the dominant kernel's HIP-event time is stable: after an idle period the first launches run
    ~9 % slower (clock ramp), so a short `--steps 20 --warmup 5` run would otherwise time the ramp.  Batches of `batch` forwards;
    stops when two consecutive batch means agree within `tol` (at least 3 batches).  Returns (launches made, last mean ms)."""
    ctx.timing_reset()
    ctx.timing(1 << slot)
    prev, n = None, 0
    for i in range(max_batches):
        for _ in range(batch):
            model(x)
        ms, cnt = ctx.timing_read(slot)
        ctx.timing_reset()
        cur = ms / max(cnt, 1)
        n += batch
        if prev is not None and i >= 2 and abs(cur - prev) <= tol * prev:
            prev = cur
            break
        prev = cur
    ctx.timing(False)
    return n, prev


def timed_steps(torch, dist, model, x, steps, warmup, world, probe_clock=False):
    """Pre-conditioning (untimed, see precondition), W untimed warm-up steps, then EXACTLY `steps` timed steps between barrier +
    synchronize on both sides.  Inside the timed region only the dominant kernel (timing slot 2) is bracketed by HIP events on
    the launch stream; the per-kernel breakdown (every launch bracketed) runs directly AFTER the timed region -- the only host
    work in between is reading back the <= 256 already-completed events of slot 2 -- so both are taken in the same clock state.
    probe_clock: the breakdown launches also stamp s_memtime / s_memrealtime inside the dominant kernel (dfa_ctx_clock_read)."""
    from dfa_amd import _lib
    ctx = _lib.Context.get(x.device)
    pre_n, _ = precondition(torch, ctx, model, x)
    for _ in range(warmup):
        model(x)
    ctx.timing_reset()
    ctx.timing(1 << 2)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = model(x)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    dominant = ctx.timing_read(2)
    ctx.timing_reset()
    ctx.timing(True)
    if probe_clock:
        ctx.set_option("clock_probe", 1)
    for _ in range(30):
        model(x)
    ctx.timing(False)
    torch.cuda.synchronize()
    clock = None
    if probe_clock:
        med, lo, hi, n = ctx.clock_read()
        ctx.set_option("clock_probe", 0)
        if n:
            clock = {"ghz": round(med, 3), "min": round(lo, 3), "max": round(hi, 3), "workgroups": n}
    slots = [ctx.timing_read(s) for s in range(4)]
    ctx.timing_reset()
    probe_ms = slots[2][0] / max(slots[2][1], 1)
    slots[2] = dominant
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=x.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, slots, out, {"precondition_launches": pre_n, "clock": clock, "probe_kernel_ms": round(probe_ms, 4)}


def host_cores():
    """Threads this process may really use: CPU affinity, clipped by the cgroup CPU quota when there is one; a box
    that exposes the whole host (hundreds of CPUs) to a one-GPU job is treated as its documented 16-core share."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(n, 16) if n > 64 else n


def cpu_baseline(torch, sd_cpu, seconds):
    """The CPU restatement of the reference's predict path (batch 32, fp32, all host threads; src/predict.py:16,100-111)."""
    from oracle import torch_ref as R
    cores = host_cores()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1234)
    xb = (torch.randn(32, F, T, generator=g) * 3.2 - 0.07).transpose(1, 2)
    R.cnn2d_forward(sd_cpu, xb[:4])            # warm the thread pool / oneDNN primitives
    n, t0 = 0, time.perf_counter()
    while True:
        R.cnn2d_forward(sd_cpu, xb)
        n += 32
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    used = torch.get_num_threads()
    # single-thread figure for normalisation (SURVEY section 8 d): one batch of 8 utterances
    torch.set_num_threads(1)
    R.cnn2d_forward(sd_cpu, xb[:1])
    t1 = time.perf_counter()
    R.cnn2d_forward(sd_cpu, xb[:8])
    one = 8 / (time.perf_counter() - t1)
    torch.set_num_threads(used)
    return {"value": round(n / el, 2), "unit": "utterances/s", "cores": used, "kind": "port",
            "single_thread_value": round(one, 2),
            "sample": f"{n} utterances ({n // 32} batches of 32, [32,321,180] fp32 strided view) in {el:.1f} s with "
                      "oracle/torch_ref.py (plain PyTorch CPU ops restating src/model.py:33-42); single-thread figure from "
                      "one batch of 8"}


def write_synthetic_pickle(torch, n, tmp):
    """features.pkl in the reference schema (README.md:45-103): DataFrame{uttid, features: FloatTensor [180, 321]}."""
    import pandas as pd
    g = torch.Generator().manual_seed(4321)
    feats = torch.randn(n, F, T, generator=g) * 3.2 - 0.07
    path = os.path.join(tmp, "features.pkl")
    pd.DataFrame({"uttid": [f"utt_{i:06d}" for i in range(n)], "features": [feats[i].clone() for i in range(n)]}).to_pickle(path)
    return path


def cpu_end_to_end(torch, sd_cpu, features_path, n):
    """src/predict.py end to end on the host (oracle/predict_ref.py): unpickle -> DataLoader(32, workers 2) -> CPU model ->
    sigmoid -> prediction.pkl."""
    from oracle import predict_ref as P
    torch.set_num_threads(host_cores())
    out = features_path.replace("features.pkl", "prediction_cpu.pkl")
    t0 = time.perf_counter()
    P.predict_end_to_end(features_path, sd_cpu, out, batch_size=32, num_workers=2)
    el = time.perf_counter() - t0
    return {"value": round(n / el, 2), "unit": "utterances/s", "seconds": round(el, 2),
            "sample": f"{n} utterances: features.pkl -> DataLoader(batch 32, num_workers 2) -> oracle/torch_ref.py -> "
                      "prediction.pkl, load and write included (src/predict.py:88-122)"}, out


def gpu_end_to_end(torch, device, model_state, features_path, n, cpu_pred_path):
    """The same file through the product: dfa_amd.predict (unpickle, stack, H2D over PCIe, kernels, D2H, pickle) and the
    flat-file ingest path of SURVEY 8(f)1 (one-time convert, then memmap -> pinned double-buffered H2D -> kernels)."""
    import numpy as np
    import pandas as pd
    from dfa_amd import ingest, predict
    tmp = os.path.dirname(features_path)
    ck = os.path.join(tmp, "cnn2d.pt")
    torch.save({"model_state": model_state}, ck)
    res = {}
    for prec in ("fp32", "bf16"):
        out = os.path.join(tmp, f"prediction_{prec}.pkl")
        argv = ["--features", features_path, "--checkpoint", ck, "--model", "cnn2d", "--out", out, "--batch-size", "256",
                "--precision", prec]
        predict.main(argv)                                   # first call pays workspace allocation / weight packing
        t0 = time.perf_counter()
        predict.main(argv)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        res[f"predict_pkl_{prec}"] = {"value": round(n / el, 1), "unit": "utterances/s", "seconds": round(el, 3)}
        if prec == "fp32" and cpu_pred_path and os.path.exists(cpu_pred_path):
            a, b = pd.read_pickle(out), pd.read_pickle(cpu_pred_path)
            res["max_abs_score_diff_vs_cpu"] = float(np.abs(a["predictions"].values - b["predictions"].values).max())
    # flat ingest: convert once (timed separately), then stream
    t0 = time.perf_counter()
    ingest.convert(features_path, os.path.join(tmp, "flat16"), None, dtype="bf16")
    res["ingest_convert_seconds"] = round(time.perf_counter() - t0, 3)
    ff = ingest.FlatFeatures(os.path.join(tmp, "flat16"))
    model = build_model(torch, device, "bf16")
    model.load_state_dict(model_state)
    flat = ff.tensor()
    predict.predict_scores(model, flat, batch_size=256, apply_sigmoid=True).cpu()
    t0 = time.perf_counter()
    reps = 4
    for _ in range(reps):
        scores = predict.predict_scores(model, flat, batch_size=256, apply_sigmoid=True).cpu()
    el = (time.perf_counter() - t0) / reps
    res["flat_ingest_bf16"] = {"value": round(n / el, 1), "unit": "utterances/s", "seconds": round(el, 4),
                               "what": "memory-mapped [N,180,321] bf16 file -> FlatBatcher (async double-buffered H2D, "
                                       "PCIe included) -> kernels -> scores on the host",
                               "host_bytes_per_utt": F * T * 2}
    # the same stream at steady state: the file tiled to >= 4096 utterances (the n-utterance sample is two batches: its rate is
    # first-batch latency + the final read-back, not the pipeline's)
    try:
        import warnings
        reps_n = max(1, -(-4096 // n))
        big_path = os.path.join(tmp, "flat16_big.npy")
        big = np.lib.format.open_memmap(big_path, mode="w+", dtype=np.uint16, shape=(reps_n * n, F, T))
        src = np.asarray(ff.array).view(np.uint16)
        for r in range(reps_n):
            big[r * n:(r + 1) * n] = src
        big.flush()
        del big
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            bt = torch.from_numpy(np.load(big_path, mmap_mode="r")).view(torch.bfloat16)
        predict.predict_scores(model, bt, batch_size=256, apply_sigmoid=True).cpu()
        t0 = time.perf_counter()
        for _ in range(2):
            sc_big = predict.predict_scores(model, bt, batch_size=256, apply_sigmoid=True).cpu()
        el = (time.perf_counter() - t0) / 2
        res["flat_ingest_bf16_steady"] = {"value": round(reps_n * n / el, 1), "unit": "utterances/s", "seconds": round(el, 4),
                                          "utterances": reps_n * n,
                                          "what": "the same path over the file tiled to >= 4096 utterances (page-cache resident memmap -> "
                                                  "double-buffered H2D over PCIe -> kernels -> scores on the host)",
                                          "scores_equal_first_tile": bool(torch.equal(sc_big[:n], scores))}
        os.remove(big_path)
    except Exception as e:  # noqa: BLE001
        res["flat_ingest_bf16_steady"] = {"error": f"{type(e).__name__}: {e}"}
    res["sample"] = f"{n} utterances, host-resident; PCIe-inclusive (never the headline value)"
    return res


def small_batch_metric(torch, device, steps=200, warmup=30):
    """The reference's predict.py default batch (32) and batch 1: a step is latency-bound there (one workgroup walk down the
    time axis per strip); the eval forward splits the time axis over workgroups (context option time_split, bit-identical
    results).  utterances/s and ms per forward, bf16 mode, with the split (default) and without."""
    from dfa_amd import _lib
    ctx = _lib.Context.get(device)
    model = build_model(torch, device, "bf16")
    g = torch.Generator().manual_seed(5)
    out = {}
    try:
        for Bs in (1, 32):
            x = (torch.randn(Bs, F, T, generator=g) * 3.2 - 0.07).to(device=device, dtype=torch.bfloat16).transpose(1, 2)
            rec = {}
            for tag, opt in (("time_split", -1), ("no_split", 0)):
                ctx.set_option("time_split", opt)
                for _ in range(warmup):
                    model(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(steps):
                    model(x)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps
                rec[tag] = {"ms": round(dt * 1e3, 4), "value": round(Bs / dt, 1)}
            out[f"batch_{Bs}"] = rec
    finally:
        ctx.set_option("time_split", -1)
    out["unit"] = "utterances/s"
    return out


TRAIN_FLOPS_PER_UTT = 9_621_849_600      # fwd 3.218 G + wgrad 3.218 G + dgrad(conv2,3) + linear 3.185 G


def train_step_metric(torch, dist, device, B, world, rank, steps=20, warmup=8):
    """BASELINE configs[2]: CNN2D training step (fwd + bwd + fused AdamW, dropout 0.2, label smoothing 0.05) in the bf16-storage
    mode, data-parallel over `world` ranks (B utterances per rank): every step ends with the SUM all-reduce of the flat 464,644-byte
    gradient buffer (RCCL over xGMI under the "nccl" backend; src/train.py:74-76 has no counterpart, the reference is
    single-device) and the fused AdamW with the 1/world scale folded in.  Same barrier + max-over-ranks protocol as the headline;
    value = GLOBAL utterances/s.  The all-reduce alone is timed separately with HIP events on the same buffer."""
    from dfa_amd.model import CNN2D
    from dfa_amd.training.train_step import NativeTrainer
    torch.manual_seed(0)
    model = CNN2D(in_features=F, dropout=0.2, precision="bf16").to(device)
    model._drop_seed = 1234 + 7919 * rank                       # ranks draw different dropout masks (train.py mixes the rank in too)
    g = torch.Generator().manual_seed(99 + rank)
    x = (torch.randn(B, F, T, generator=g) * 3.2 - 0.07).to(device=device, dtype=torch.bfloat16).transpose(1, 2)
    y = (torch.rand(B, generator=g) > 0.5).float().to(device)
    tr = NativeTrainer(model, label_smoothing=0.05)
    for _ in range(warmup):
        tr.step(x, y)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(x, y)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ar_ms = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        buf = torch.zeros_like(tr.flat_g)
        for _ in range(5):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        e0.record()
        for _ in range(50):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        ar_ms = e0.elapsed_time(e1) / 50
    dt /= steps
    ach = B * TRAIN_FLOPS_PER_UTT / dt / 1e12                   # per GPU
    traffic, src = pmc_step_traffic("train_step")
    out = {"value": round(world * B / dt, 1), "unit": "utterances/s", "ms_per_step": round(dt * 1e3, 3), "dtype": "bf16",
           "batch_per_gpu": B, "global_batch": world * B, "n_gpus": world, "loss_rank0": round(float(loss.item()), 4),
           "roofline": {"bound": "mfma", "scope": "whole step (forward + backward + all-reduce + AdamW), algorithmic FLOPs / wall "
                                                   "time, per GPU",
                        "achieved": round(ach, 2), "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_TFLOPS["bf16"], 4), "flops_per_utt": TRAIN_FLOPS_PER_UTT,
                        "traffic": traffic if B == B_PER_GPU else None, "traffic_source": src},
           "what": f"fwd + bwd + one flat-gradient all-reduce + fused AdamW, dropout 0.2, label smoothing 0.05 "
                   f"(src/train.py:71-76), data-parallel x{world}"}
    if world > 1:
        out["allreduce"] = {"ms": round(ar_ms, 4), "bytes": tr.flat_g.numel() * 4, "backend": dist.get_backend(),
                            "what": "SUM all-reduce of the flat fp32 gradient alone (HIP events, mean of 50)"}
    return out


CNN1D_BYTES_PER_UTT = 231_124            # read x once (fp32) + 4 B out
CNN1D_FLOPS_PER_UTT = 30_816_256
CAE_FLOPS_PER_UTT = 1_792_021_760        # CAE forward (+ MSE)
CAE_TRAIN_FLOPS_PER_UTT = 3 * CAE_FLOPS_PER_UTT - 2 * 16_640_640   # fwd + wgrad + dgrad (no data gradient through encoder block 1)


def other_models_metric(torch, device, B, steps=20, warmup=5):
    """Secondary paths of SURVEY section 8 at the same batch (rank 0, N = 1 only), each with the roofline section 8(d) assigns it:
    CNN1D forward (a3: HBM-bound, 231,124 B per utterance), the auto-encoder's anomaly score with the z-score and per-sample MSE
    fused in (a4/a5/a13: MFMA-bound, 1.792 GFLOP per utterance) and the auto-encoder training step."""
    from dfa_amd import _lib
    from dfa_amd.model_cae import ConvAutoencoder
    from dfa_amd.model_cnn1d import CNN1D
    ctx = _lib.Context.get(device)
    g = torch.Generator().manual_seed(77)
    stored = (torch.randn(B, F, T, generator=g) * 3.2 - 0.07).to(device)
    x = stored.transpose(1, 2)

    def rate(fn, slots=()):
        for _ in range(warmup + 10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        rec = {"value": round(B / dt, 1), "unit": "utterances/s", "ms_per_step": round(dt * 1e3, 4)}
        if slots:                                           # per-kernel HIP-event times of the same launches, after the timed loop
            ctx.timing_reset()
            ctx.timing(True)
            for _ in range(20):
                fn()
            ctx.timing(False)
            torch.cuda.synchronize()
            rec["kernel_ms"] = {}
            for name, sl in slots:
                ms, n = ctx.timing_read(sl)
                if n:
                    rec["kernel_ms"][name] = round(ms / n, 4)
            ctx.timing_reset()
        return rec

    torch.manual_seed(0)
    out = {}
    m1 = CNN1D(in_features=F).to(device).eval()
    r = rate(lambda: m1(x), slots=(("cnn1d_fused", 4), ("conv2", 5), ("conv3", 6), ("linear", 7)))
    k_ms = sum(r["kernel_ms"].values()) if r.get("kernel_ms") else r["ms_per_step"]
    gbs = B * CNN1D_BYTES_PER_UTT / (k_ms * 1e-3) / 1e9
    traffic, src = pmc_step_traffic("cnn1d_fwd")
    r["roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4),
                     "bytes_per_utt": CNN1D_BYTES_PER_UTT, "kernel_ms": round(k_ms, 4),
                     "traffic": traffic if B == B_PER_GPU else None, "traffic_source": src}
    r["dtype"] = "bf16x3 (hi + lo bf16 operands, three MFMAs per product, fp32 accumulate; fp32 features in, logits within 1e-5 of fp32)"
    out["cnn1d_fwd"] = r
    mean, std = torch.zeros(F, device=device), torch.ones(F, device=device)
    cae = ConvAutoencoder(precision="bf16").to(device).eval()
    x16 = x.to(torch.bfloat16)
    r = rate(lambda: cae.score(x16, mean, std),
             slots=(("enc1", 8), ("enc2", 9), ("enc3", 10), ("enc4", 11), ("decoder_fused_mse", 12), ("dec2", 13), ("dec3", 14), ("dec4_mse", 15)))
    tf = B * CAE_FLOPS_PER_UTT / (r["ms_per_step"] * 1e-3) / 1e12
    traffic, src = pmc_step_traffic("cae_score")
    r["roofline"] = {"bound": "mfma", "scope": "whole score (encoder + decoder + MSE), algorithmic FLOPs / wall time",
                     "achieved": round(tf, 2), "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s",
                     "frac": round(tf / PEAK_TFLOPS["bf16"], 4), "flops_per_utt": CAE_FLOPS_PER_UTT,
                     "traffic": traffic if B == B_PER_GPU else None, "traffic_source": src}
    out["cae_score_bf16"] = r
    try:
        out["cae_train_step_bf16"] = cae_train_metric(torch, device, B)
    except Exception as e:                                   # secondary leg: report, never lose the headline line
        out["cae_train_step_bf16"] = {"error": f"{type(e).__name__}: {e}"}
    try:
        out["cnn1d_train_step"] = cnn1d_train_metric(torch, device, B)
    except Exception as e:  # noqa: BLE001
        out["cnn1d_train_step"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def cae_train_metric(torch, device, B, steps=10, warmup=4):
    """Auto-encoder training step (src/train_cae.py:58-82: recon = model(x); MSELoss(recon, x); backward; AdamW lr 1e-4 wd 1e-4) in
    the bf16-storage mode on the flat-buffer trainer."""
    from dfa_amd.model_cae import ConvAutoencoder
    from dfa_amd.training.train_step import make_cae_trainer
    torch.manual_seed(0)
    model = ConvAutoencoder(precision="bf16").to(device).train()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, T, F, generator=g).to(device=device, dtype=torch.bfloat16)
    tr = make_cae_trainer(model, lr=1e-4, weight_decay=1e-4)
    for _ in range(warmup):
        tr.step(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    tf = B * CAE_TRAIN_FLOPS_PER_UTT / dt / 1e12
    traffic, src = pmc_step_traffic("cae_train_step")
    return {"value": round(B / dt, 1), "unit": "utterances/s", "ms_per_step": round(dt * 1e3, 3), "loss": round(float(loss), 5),
            "trainer": type(tr).__name__,
            "roofline": {"bound": "mfma", "scope": "whole step, algorithmic FLOPs / wall time", "achieved": round(tf, 2),
                         "peak": PEAK_TFLOPS["bf16"], "unit": "TFLOP/s", "frac": round(tf / PEAK_TFLOPS["bf16"], 4),
                         "flops_per_utt": CAE_TRAIN_FLOPS_PER_UTT, "traffic": traffic if B == B_PER_GPU else None,
                         "traffic_source": src}}


def cnn1d_train_metric(torch, device, B, steps=30, warmup=6):
    """CNN1D training step (src/train.py:71-76 with --model cnn1d: BCE on smoothed labels, backward, AdamW) on the all-C-ABI
    trainer, fp32 features in the stored [B, F, T] layout: convolutions and weight gradients on the matrix cores with three bf16
    terms per operand (fp32-grade sums).  Bound by HBM in SURVEY's table: x is read twice (layer-1
    convolution, layer-1 weight gradient) and every activation / gradient tensor written once and read twice."""
    from dfa_amd.model_cnn1d import CNN1D
    from dfa_amd.training.train_step import NativeTrainer
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(6)
    x = (torch.randn(B, F, T, generator=g) * 3.2 - 0.07).to(device).transpose(1, 2)
    y = (torch.rand(B, generator=g) > 0.5).float().to(device)
    tr = NativeTrainer(CNN1D(in_features=F, dropout=0.2).to(device), label_smoothing=0.05)
    for _ in range(warmup):
        tr.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    # algorithmic bytes per utterance: x twice; z1..z3, h1, h2 (fp32 [C][T]) written once and read twice; dz3..dz1, dh2, dh1 likewise
    elems = (32 + 64 + 128) + (32 + 64) + (128 + 64 + 32) + (64 + 32)
    bytes_per_utt = 2 * F * T * 4 + 3 * elems * T * 4
    gbs = B * bytes_per_utt / dt / 1e9
    traffic, src = pmc_step_traffic("cnn1d_train_step")
    return {"value": round(B / dt, 1), "unit": "utterances/s", "ms_per_step": round(dt * 1e3, 4), "loss": round(float(loss), 5),
            "trainer": "NativeTrainer(kind='cnn1d')", "dtype": "fp32 (three bf16 terms per operand on the matrix cores)",
            "roofline": {"bound": "hbm", "scope": "whole step, algorithmic bytes / wall time", "achieved": round(gbs, 1), "peak": 8000.0,
                         "unit": "GB/s", "frac": round(gbs / 8000.0, 4), "bytes_per_utt": bytes_per_utt,
                         "traffic": traffic if B == B_PER_GPU else None, "traffic_source": src}}


def spawn_ranks(args):
    """`python bench.py --gpus N` by itself: start the N ranks as a CHILD torch.distributed.run (one process per GPU,
    RCCL) before this process has made any GPU call, relay its output and exit with its code.  Never an exec."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback in dfa_amd)")
    # DFA_BENCH_SHARE_GPU=1 is a rehearsal switch for 1-GPU boxes: every rank uses device 0 and the (timing-only)
    # collectives run over gloo, so the N > 1 code path can be exercised without N GPUs.  Never set for real runs.
    share = os.environ.get("DFA_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=device)   # nccl == RCCL on ROCm

    B = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    stored = (torch.randn(B, F, T, generator=g) * 3.2 - 0.07)        # stored layout [B,F,T] (src/dataset.py:52)
    x32 = stored.to(device).transpose(1, 2)                           # the strided [B,T,F] view the harness feeds
    x16 = stored.to(device=device, dtype=torch.bfloat16).transpose(1, 2)

    results = {}
    for prec, x in (("bf16", x16), ("bf16x3", x32), ("fp32", x32)):
        model = build_model(torch, device, prec)
        dt, slots, out, extra = timed_steps(torch, dist, model, x, args.steps, args.warmup, world, probe_clock=(prec == "bf16"))
        if not torch.isfinite(out).all():
            raise SystemExit(f"non-finite logits in {prec} mode")
        ms3, n3 = slots[2]
        k_ms = ms3 / max(n3, 1)
        ach = BLOCK3_FLOPS_PER_UTT * B / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        traffic, traffic_src = pmc_kernel_traffic(PMC_TAGS[prec])
        results[prec] = {
            "value": world * B * args.steps / dt,
            "ms_per_step": dt / args.steps * 1e3,
            "roofline": {"bound": "mfma", "kernel": BLOCK3_KERNEL[prec] + " (CNN2D block 3, 64->128, +BN+ReLU+mean_T)",
                         "achieved": round(ach, 2), "peak": round(PEAK_TFLOPS[prec], 1), "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_TFLOPS[prec], 4),
                         "traffic": traffic if B == B_PER_GPU else None, "traffic_source": traffic_src,
                         "kernel_ms": round(k_ms, 4), "launches_timed": n3,
                         "precondition_launches": extra["precondition_launches"]},
            "kernel_ms": {name: round(ms / max(n, 1), 4) for name, (ms, n) in
                          zip(("conv1", "block2_mfma_or_fused_blocks12", "block3_mfma", "linear"), slots)},
            "logits_sample": [round(float(v), 6) for v in out[:3, 0].float().cpu()],
        }
        if extra["clock"]:
            # the 2500 TFLOP/s peak is quoted at 2.4 GHz; under MFMA load the chip holds less (MI355X_MICROARCH.md, DVFS give-back):
            # clock_ghz is measured INSIDE the dominant kernel (s_memtime / s_memrealtime around its main loop, median over 1024
            # workgroups) in the launches that directly follow the timed region, so box spread and kernel regressions separate
            rf = results[prec]["roofline"]
            rf["clock_ghz"] = extra["clock"]["ghz"]
            rf["clock_ghz_min_max"] = [extra["clock"]["min"], extra["clock"]["max"]]
            rf["frac_at_held_clock"] = round(ach / (PEAK_TFLOPS[prec] * extra["clock"]["ghz"] / 2.4), 4)
            rf["probe_kernel_ms"] = extra["probe_kernel_ms"]
        sd_cpu = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        del model

    if rank == 0:
        r16, r32, rx3 = results["bf16"], results["fp32"], results["bf16x3"]
        line = {
            "metric": METRIC, "value": round(r16["value"], 1), "unit": "utterances/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(r16["ms_per_step"], 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: CNN2D eval forward, synthetic [256,321,180] per GPU, bf16 "
                                   "storage / fp32 accumulate, strided view of stored [B,180,321]",
                       "batch_per_gpu": B, "T": T, "F": F, "parallelism": f"utterance-sharded x{world}, no collective"},
            "flops_per_utt": FLOPS_PER_UTT,
            "achieved_tflops_whole_path": round(r16["value"] / world * FLOPS_PER_UTT / 1e12, 2),
            "roofline": r16["roofline"],
            "kernel_ms": r16["kernel_ms"],
            "parity_fast": {"value": round(rx3["value"], 1), "unit": "utterances/s", "dtype": "bf16x3",
                            "what": "DFA_PREC_BF16X3: hi + lo bf16 operands, three bf16 MFMAs per product, fp32 accumulate; "
                                    "logits within 1e-4 of the reference and identical EER on the N=2000 set (GPU tests); "
                                    "roofline peak = bf16 MFMA peak / 3 (algorithmic FLOPs)",
                            "ms_per_step": round(rx3["ms_per_step"], 4), "roofline": rx3["roofline"],
                            "kernel_ms": rx3["kernel_ms"],
                            "max_abs_logit_diff_vs_fp32": round(max(abs(a - b) for a, b in
                                                                    zip(rx3["logits_sample"], r32["logits_sample"])), 6)},
            "fp32_parity": {"value": round(r32["value"], 1), "unit": "utterances/s", "dtype": "f32",
                            "ms_per_step": round(r32["ms_per_step"], 4), "roofline": r32["roofline"],
                            "kernel_ms": r32["kernel_ms"],
                            "max_abs_logit_diff_vs_bf16": round(max(abs(a - b) for a, b in
                                                                    zip(r16["logits_sample"], r32["logits_sample"])), 5)},
        }
        if world == 1 and args.small_batch:   # opt-in: its B = 1 / 32 launches would mix into the per-kernel averages of a rocprofv3 run
            line["small_batch"] = small_batch_metric(torch, device)
    # BASELINE configs[2]: the data-parallel training step runs on EVERY rank (its all-reduce is a collective)
    train = None if args.no_train else train_step_metric(torch, dist, device, B, world, rank)
    if rank == 0:
        if train is not None:
            line["train_step"] = train
        if world == 1 and not args.no_other_models:
            line["other_models"] = other_models_metric(torch, device, B)
        if world == 1 and not args.no_cpu_baseline:
            import tempfile
            line["cpu_baseline"] = cpu_baseline(torch, sd_cpu, args.cpu_seconds)
            with tempfile.TemporaryDirectory() as tmp:
                n_e2e = args.e2e_utts
                fpath = write_synthetic_pickle(torch, n_e2e, tmp)
                line["cpu_baseline"]["end_to_end"], cpu_pred = cpu_end_to_end(torch, sd_cpu, fpath, n_e2e)
                line["end_to_end"] = gpu_end_to_end(torch, device, sd_cpu, fpath, n_e2e, cpu_pred)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
