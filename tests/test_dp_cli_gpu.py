"""Data-parallel training drivers on the GPU box: two ranks share the one test GPU (gloo carries the collectives: RCCL
refuses two ranks on one device), the utterance counts are NOT multiples of world x batch (the case that used to hang),
every model family goes through its flat-buffer trainer, and the "nccl" (RCCL) branch runs as a one-rank group."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _make_pickles(tmp, n, tag, seed, T=48):
    import pandas as pd
    g = torch.Generator().manual_seed(seed)
    pattern = torch.outer(torch.sin(torch.arange(180) / 5.0), torch.cos(torch.arange(T) / 17.0))
    labels = (torch.rand(n, generator=g) > 0.5).long()
    feats = [torch.randn(180, T, generator=g) + 3.0 * (2 * labels[i] - 1) * pattern for i in range(n)]
    ids = [f"{tag}{i:04d}" for i in range(n)]
    fp, lp = os.path.join(tmp, f"{tag}_features.pkl"), os.path.join(tmp, f"{tag}_labels.pkl")
    pd.DataFrame({"uttid": ids, "features": feats}).to_pickle(fp)
    pd.DataFrame({"uttid": ids, "label": labels.numpy()}).to_pickle(lp)
    return fp, lp


def _cli_worker(rank, world, port, tmp, which):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", DFA_DIST_BACKEND="gloo")
    import torch.distributed as dist
    import dfa_amd  # noqa: F401
    common = ["--train-features", os.path.join(tmp, "tr_features.pkl"), "--train-labels", os.path.join(tmp, "tr_labels.pkl"),
              "--dev-features", os.path.join(tmp, "dv_features.pkl"), "--dev-labels", os.path.join(tmp, "dv_labels.pkl"),
              "--batch-size", "16", "--num-workers", "0", "--checkpoint-dir", tmp]
    if which == "cae":
        from dfa_amd import train_cae as TC
        TC.main(common + ["--epochs", "2", "--run-name", "cae_dp", "--seed", "3"])
    else:
        from dfa_amd import train as T
        T.main(common + ["--model", which, "--epochs", "3", "--run-name", f"{which}_dp", "--seed", "1", "--label-smoothing", "0.05",
                         "--lr-scheduler", "plateau", "--lr-scheduler-patience", "0", "--early-stop", "3", "--time-shift"])
    # every rank ends with the same parameters and BatchNorm statistics
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(tmp, f"{which}_done{rank}"), "w").write("ok")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("which", ["cnn2d", "cnn1d", "cae"])
def test_data_parallel_cli_two_ranks_ragged_counts(tmp_path, which):
    import torch.multiprocessing as mp
    from dfa_amd.training import load_checkpoint
    tmp = str(tmp_path)
    _make_pickles(tmp, 50, "tr", 0)          # 50 = 1 global batch of 32 + 18: ceil-sharding gave 2 and 2 steps of unequal size;
    _make_pickles(tmp, 21, "dv", 1)          # 21 dev utterances: shards of 11 and 10
    port = 29700 + (os.getpid() % 1000)
    mp.spawn(_cli_worker, args=(2, port, tmp, which), nprocs=2, join=True)
    assert all(os.path.exists(os.path.join(tmp, f"{which}_done{r}")) for r in range(2))
    name = "cae" if which == "cae" else which
    run = "cae_dp" if which == "cae" else f"{which}_dp"
    blob = load_checkpoint(os.path.join(tmp, run, f"{name}_last.pt"))
    assert set(blob) >= {"model_state", "optimizer_state", "epoch", "config", "scheduler_state"}
    st = blob["optimizer_state"]
    steps = {int(v["step"]) for v in st["state"].values()}
    assert len(steps) == 1 and steps.pop() >= 2                      # the fused optimiser's state in torch's format
    assert st["param_groups"][0]["lr"] > 0
    if which != "cae":
        assert blob["scheduler_state"]["patience"] == 0


def test_rccl_branch_single_rank_group():
    """backend "nccl" is RCCL on ROCm: run the plumbing's collectives through it as a one-rank group (the test box has
    one GPU), so the branch that the 8-GPU job takes has at least executed on hardware."""
    import torch.distributed as dist
    from dfa_amd import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90), RANK="0", WORLD_SIZE="1")
    os.environ.pop("DFA_DIST_BACKEND", None)
    try:
        rank, world = D.init(device=torch.device("cuda", 0), force_group=True)
        assert (rank, world) == (0, 1) and dist.get_backend() == "nccl"
        g = torch.arange(116161, dtype=torch.float32, device="cuda")         # the CNN2D flat gradient size
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        assert torch.equal(g, torch.arange(116161, dtype=torch.float32, device="cuda"))
        p = torch.ones(7, device="cuda")
        dist.broadcast(p, src=0)
        assert D.mean_scalar(2.5, torch.device("cuda", 0)) == 2.5
        t = [torch.full((3,), 2.0, device="cuda")]
        dist.all_reduce(t[0]); torch.cuda.synchronize()
        assert D.gather_scores([1.0, 2.0]).tolist() == [1.0, 2.0]
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _syncbn_make(which, prec, dev, sync):
    """(model, trainer, step(x, y) -> loss) for one of the three model families, dropout off, same seed everywhere"""
    from dfa_amd.training.train_step import CaeNativeTrainer, NativeTrainer
    torch.manual_seed(0)
    if which == "cae":
        from dfa_amd.model_cae import ConvAutoencoder
        model = ConvAutoencoder(precision=prec).to(dev)
        tr = CaeNativeTrainer(model, lr=1e-4, weight_decay=1e-4, sync_bn=sync)
        return model, tr, (lambda x, y: tr.step(x))
    if which == "cnn1d":
        from dfa_amd.model_cnn1d import CNN1D
        model = CNN1D(in_features=180, dropout=0.0).to(dev)
    else:
        from dfa_amd.model import CNN2D
        model = CNN2D(in_features=180, dropout=0.0, precision=prec).to(dev)
    tr = NativeTrainer(model, lr=1e-3, weight_decay=0.01, label_smoothing=0.05, sync_bn=sync)
    return model, tr, (lambda x, y: tr.step(x, y))


def _syncbn_worker(rank, world, port, tmp, which, prec, sync):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", DFA_DIST_BACKEND="gloo")
    import torch.distributed as dist
    import dfa_amd  # noqa: F401
    from dfa_amd import distributed as D
    dev = torch.device("cuda", 0)
    D.init(device=dev)
    x, y = _syncbn_batch(which)
    per = x.shape[0] // world
    model, tr, step = _syncbn_make(which, prec, dev, sync)
    loss = step(x[rank * per:(rank + 1) * per].to(dev), y[rank * per:(rank + 1) * per].to(dev))
    torch.cuda.synchronize()
    out = {"flat_g": tr.flat_g.detach().cpu() / world, "loss": float(loss),
           "stats": {k: v.detach().cpu().clone() for k, v in model.state_dict().items() if "running" in k}}
    torch.save(out, os.path.join(tmp, f"syncbn_{which}_{prec}_{int(sync)}_{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _syncbn_batch(which="cnn2d"):
    g = torch.Generator().manual_seed(11)
    B, T = 8, (96 if which == "cae" else 66)
    stored = torch.randn(B, 180, T, generator=g) * 3.0 + torch.linspace(-2, 2, B).view(B, 1, 1)     # utterances differ in level: the halves have different statistics
    return stored.transpose(1, 2), (torch.rand(B, generator=g) > 0.5).float()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("which,prec", [("cnn2d", "fp32"), ("cnn2d", "bf16"), ("cnn1d", "fp32"), ("cae", "fp32"), ("cae", "bf16")])
def test_sync_bn_two_ranks_train_like_one_rank_with_the_whole_batch(tmp_path, which, prec):
    """"SyncBN ... makes N GPUs x B equal to one GPU x N*B up to summation order -- recommended for
    testability".  With `sync_bn=True` on the native trainers (dfa_ctx_set_bn_sync: every BatchNorm layer's per-channel sums are
    added over the ranks between their reduction and their use, two 2C-float all-reduces per layer and step) two ranks with half
    the batch each end one step with the SAME running statistics as one rank with the whole batch and with an averaged gradient
    equal to the whole-batch gradient (src/train.py:71-76 / src/train_cae.py:58-82) -- up to summation order, i.e. up to the
    ReLU-flip noise of tests/test_train_shapes_gpu.py; without it (DistributedDataParallel's local statistics, the default) the
    running statistics of these deliberately different halves do not match, which is the control that the hook is what does it.
    All three model families; the bf16-storage modes to their storage noise."""
    import torch.multiprocessing as mp
    tmp = str(tmp_path)
    port = 30900 + (os.getpid() % 1000)
    for sync in (True, False):
        mp.spawn(_syncbn_worker, args=(2, port + int(sync), tmp, which, prec, sync), nprocs=2, join=True)
    dev = torch.device("cuda", 0)
    x, y = _syncbn_batch(which)
    model, tr, step = _syncbn_make(which, prec, dev, False)
    loss = step(x.to(dev), y.to(dev))
    want_g = tr.flat_g.detach().cpu()
    want_stats = {k: v.detach().cpu() for k, v in model.state_dict().items() if "running" in k}
    r0, r1 = (torch.load(os.path.join(tmp, f"syncbn_{which}_{prec}_1_{r}.pt")) for r in range(2))
    assert torch.equal(r0["flat_g"], r1["flat_g"])                       # the ranks agree bit for bit (same all-reduced sums)
    for k, v in want_stats.items():
        assert torch.equal(r0["stats"][k], r1["stats"][k]), k
        tol = 1e-5 if prec == "fp32" else 2e-3
        assert float((r0["stats"][k] - v).abs().max()) <= tol * max(1.0, float(v.abs().max())), (k, float((r0["stats"][k] - v).abs().max()))
    rel = float((r0["flat_g"] - want_g).norm() / want_g.norm())
    assert rel <= (3e-3 if prec == "fp32" else (1e-1 if which == "cae" else 3e-2)), rel
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - float(loss)) <= 1e-4 * max(1.0, abs(float(loss)))   # mean of the half-batch losses
    # control: local statistics (the default) -- the deliberately different halves give other running statistics
    c0 = torch.load(os.path.join(tmp, f"syncbn_{which}_{prec}_0_0.pt"))
    worst = max(float((c0["stats"][k] - v).abs().max()) / max(1.0, float(v.abs().max())) for k, v in want_stats.items())
    assert worst > 1e-3, worst
    print(f"sync_bn {which} {prec}: averaged gradient vs whole-batch gradient, relative L2 {rel:.2e}; control stats mismatch {worst:.2e}")
