#!/usr/bin/env python3
"""Headline benchmark: utterances/s of the CNN2D eval forward (BASELINE.json configs[1]: synthetic [256,321,180],
bf16 storage / fp32 accumulate) on N MI355X GPUs of one node, one process per GPU.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of 256 utterances per GPU, inputs already resident in HBM.
Inference shards by utterance with no data-path collective, so N GPUs run N independent
shards ("scaling": "weak"); the only collectives are the timing barrier and the max-over-ranks of the elapsed time.
Rank 0 prints ONE JSON line.  Extra objects on it:
  roofline     -- the dominant kernel (block-3 MFMA conv): algorithmic FLOPs per launch / average launch duration
                  measured with HIP events on the launch stream during the timed steps, vs the dense MFMA peak
  cpu_baseline -- the CPU restatement of the reference path (oracle/torch_ref.py) timed on this host, rank 0, N=1
  fp32_parity  -- the same workload in the exact-fp32 parity mode (logits within 1e-4 of the reference)
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
BLOCK3_KERNEL = {"bf16": "conv3_m16_meant_kernel", "fp32": "conv3x3_mfma_kernel<float,64,...>"}
BLOCK3_FLOPS_PER_UTT = 2 * 1_061_683_200  # Conv2d 64->128 on (80,180): the dominant kernel (66 % of the FLOPs)
PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA peaks, MI355X_MICROARCH.md "Chip-level parameters"


def pmc_traffic(prec):
    """HBM bytes per launch of the block-3 kernel from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json;
    counters cannot be read from inside this process).  None if the summary is absent."""
    try:
        blob = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
    except (OSError, ValueError):
        return None
    tag = "conv3_m16_meant_kernel" if prec == "bf16" else "conv3x3_mfma_kernel<float, 64, 4"
    for name, rec in blob.get("kernels", {}).items():
        if tag in name:
            return rec.get("hbm_bytes_per_launch")
    return None


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    # defaults: 200 timed steps after 50 warm-up steps (0.2 s of GPU time).  The first ~20 launches after an idle period
    # run ~9 % slower (clock ramp): 20/5 reads 344 k utt/s where 100/20 and 400/100 read 374-377 k on the same box.
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=50)
    p.add_argument("--batch", type=int, default=B_PER_GPU, help="utterances per GPU per step")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="time budget of the CPU baseline sample")
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


def timed_steps(torch, dist, model, x, steps, warmup, world):
    from dfa_amd import _lib
    ctx = _lib.Context.get(x.device)
    # per-kernel breakdown from an untimed pass (every launch bracketed by events) BEFORE the warm-up steps, so that the
    # warm-up runs right up to the barrier of the timed region; inside the timed region only the dominant kernel (slot 2)
    # is bracketed
    for _ in range(10):                     # clock ramp / first-touch launches stay out of the per-kernel averages
        model(x)
    torch.cuda.synchronize()
    ctx.timing_reset()
    ctx.timing(True)
    for _ in range(20):
        model(x)
    ctx.timing(False)
    torch.cuda.synchronize()
    breakdown = [ctx.timing_read(s) for s in range(4)]
    ctx.timing_reset()
    ctx.timing(1 << 2)
    for _ in range(warmup):
        model(x)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ctx.timing_reset()                      # the roofline's launch durations cover the timed steps only
    t0 = time.perf_counter()
    for _ in range(steps):
        out = model(x)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ctx.timing(False)
    slots = list(breakdown)
    slots[2] = ctx.timing_read(2)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=x.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, slots, out


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


def train_step_metric(torch, device, B, steps=20, warmup=5):
    """Secondary metric (BASELINE configs[2]): CNN2D training step (fwd + bwd + fused AdamW, dropout 0.2, label
    smoothing 0.05) in the bf16-storage mode, utterances/s on this rank."""
    from dfa_amd.model import CNN2D
    from dfa_amd.training.train_step import NativeTrainer
    torch.manual_seed(0)
    model = CNN2D(in_features=F, dropout=0.2, precision="bf16").to(device)
    g = torch.Generator().manual_seed(99)
    x = (torch.randn(B, F, T, generator=g) * 3.2 - 0.07).to(device=device, dtype=torch.bfloat16).transpose(1, 2)
    y = (torch.rand(B, generator=g) > 0.5).float().to(device)
    tr = NativeTrainer(model, label_smoothing=0.05)
    for _ in range(warmup):
        tr.step(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(B / dt, 1), "unit": "utterances/s", "ms_per_step": round(dt * 1e3, 3), "dtype": "bf16",
            "batch_per_gpu": B, "loss": round(float(loss.item()), 4),
            "what": "fwd + bwd + fused AdamW, dropout 0.2, label smoothing 0.05 (src/train.py:71-76), 1 GPU"}


def other_models_metric(torch, device, B, steps=20, warmup=5):
    """Secondary paths of SURVEY section 8 at the same batch (rank 0, N = 1 only): CNN1D forward (a3), the auto-encoder's
    anomaly score with the z-score and per-sample MSE fused in (a4/a5/a13), and the auto-encoder training step."""
    from dfa_amd.model_cae import ConvAutoencoder
    from dfa_amd.model_cnn1d import CNN1D
    g = torch.Generator().manual_seed(77)
    stored = (torch.randn(B, F, T, generator=g) * 3.2 - 0.07).to(device)
    x = stored.transpose(1, 2)

    def rate(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        return {"value": round(B / dt, 1), "unit": "utterances/s", "ms_per_step": round(dt * 1e3, 3)}

    torch.manual_seed(0)
    out = {}
    m1 = CNN1D(in_features=F).to(device).eval()
    out["cnn1d_fwd_fp32"] = rate(lambda: m1(x))
    mean, std = torch.zeros(F, device=device), torch.ones(F, device=device)
    cae = ConvAutoencoder(precision="bf16").to(device).eval()
    x16 = x.to(torch.bfloat16)
    out["cae_score_bf16"] = rate(lambda: cae.score(x16, mean, std))
    return out


def main():
    args = parse_args()
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
    for prec, x in (("bf16", x16), ("fp32", x32)):
        model = build_model(torch, device, prec)
        dt, slots, out = timed_steps(torch, dist, model, x, args.steps, args.warmup, world)
        if not torch.isfinite(out).all():
            raise SystemExit(f"non-finite logits in {prec} mode")
        ms3, n3 = slots[2]
        k_ms = ms3 / max(n3, 1)
        ach = BLOCK3_FLOPS_PER_UTT * B / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
        results[prec] = {
            "value": world * B * args.steps / dt,
            "ms_per_step": dt / args.steps * 1e3,
            "roofline": {"bound": "mfma", "kernel": BLOCK3_KERNEL[prec] + " (CNN2D block 3, 64->128, +BN+ReLU+mean_T)",
                         "achieved": round(ach, 2), "peak": PEAK_TFLOPS[prec], "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_TFLOPS[prec], 4),
                         "traffic": pmc_traffic(prec) if B == B_PER_GPU else None,
                         "kernel_ms": round(k_ms, 4), "launches_timed": n3},
            "kernel_ms": {name: round(ms / max(n, 1), 4) for name, (ms, n) in
                          zip(("conv1", "block2_mfma_or_fused_blocks12", "block3_mfma", "linear"), slots)},
            "logits_sample": [round(float(v), 4) for v in out[:3, 0].float().cpu()],
        }
        sd_cpu = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        del model

    if rank == 0:
        r16, r32 = results["bf16"], results["fp32"]
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
            "fp32_parity": {"value": round(r32["value"], 1), "unit": "utterances/s", "dtype": "f32",
                            "ms_per_step": round(r32["ms_per_step"], 4), "roofline": r32["roofline"],
                            "kernel_ms": r32["kernel_ms"],
                            "max_abs_logit_diff_vs_bf16": round(max(abs(a - b) for a, b in
                                                                    zip(r16["logits_sample"], r32["logits_sample"])), 5)},
        }
        if world == 1:
            line["train_step"] = train_step_metric(torch, device, B)
            line["other_models"] = other_models_metric(torch, device, B)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(torch, sd_cpu, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
