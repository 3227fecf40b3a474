#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/ by running the REFERENCE itself.

Run only in the build container, where the reference checkout is mounted read-only at
/root/reference:      python tests/golden/make_golden.py
The reference's modules are imported unmodified (sys.path -> /root/reference/src and
/root/reference/scripts), driven on CPU with fixed seeds, and only *data* (inputs, weights,
expected outputs) is written out.  Nothing here ships to the GPU box except the .npz files.

Fixtures (all float32 unless noted):
  cnn2d_eval.npz   state_dict + inputs (stored [B,F,T] layout) + logits/embedding at T in {321,64,7},
                   per-layer activations for a T=16 case
  cnn1d_eval.npz   same for CNN1D
  cae_eval.npz     state_dict + z-scored inputs + recon/latent/per-sample MSE at T in {321,64,70}
  cnn2d_train.npz  dropout=0 train step(s): loss, dlogits, per-parameter grads, BN running stats,
                   parameters after 1 and 3 AdamW steps, with label smoothing 0 and 0.05
  cnn1d_train.npz / cae_train.npz  same idea (BCE / MSE)
  host.npz         calculate_eer known answers, FeatureNormalizer, normalise_scores, alpha sweep,
                   ensemble mean, confusion_at_threshold
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
sys.path.insert(0, os.path.join(REF, "scripts"))
OUT = os.path.dirname(os.path.abspath(__file__))

from model import CNN2D  # noqa: E402  (reference)
from model_cnn1d import CNN1D  # noqa: E402
from model_cae import ConvAutoencoder  # noqa: E402
import evaluation as ref_eval_src  # noqa: E402  src/evaluation.py
from dataset_cae import FeatureNormalizer  # noqa: E402
from hybrid_ensemble import normalise_scores  # noqa: E402

import importlib.util  # noqa: E402
_spec = importlib.util.spec_from_file_location("ref_scripts_evaluation", os.path.join(REF, "scripts", "evaluation.py"))
ref_eval_scripts = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(ref_eval_scripts)

torch.set_num_threads(8)


def lfcc_like(gen, *shape):
    """N(-0.07, 3.2^2) clipped to [-61, 87]: the LFCC statistics the reference reports
    (results/archive/20260206_final_prep/model_prediction_report.md:24-29)."""
    return (torch.randn(*shape, generator=gen) * 3.2 - 0.07).clamp_(-61.0, 87.0)


def sd_np(model):
    return {k: v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def randomise_bn(model, gen):
    """Make BN affine params non-trivial (fresh init is gamma=1, beta=0)."""
    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            with torch.no_grad():
                m.weight.copy_(0.5 + torch.rand(m.weight.shape, generator=gen))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=gen))


def warm_up(model, batches):
    """A few train-mode passes so BN running stats are not (0, 1)."""
    model.train()
    with torch.no_grad():
        for xb in batches:
            model(xb)
    model.eval()


# ----------------------------------------------------------------------------------------- CNN2D / CNN1D eval
def make_classifier_eval(name, cls, scale):
    gen = torch.Generator().manual_seed(1234)
    torch.manual_seed(0)
    model = cls(in_features=180, dropout=0.0)
    randomise_bn(model, gen)
    warm_up(model, [lfcc_like(gen, 4, 180, 321).transpose(1, 2) for _ in range(2)])
    with torch.no_grad():
        model.classifier.weight.mul_(scale)  # fresh-init logits are ~0.02: make 1e-4 meaningful
        model.classifier.bias.fill_(0.3)
    out = {"sd." + k: v for k, v in sd_np(model).items()}
    with torch.no_grad():
        for tag, (B, T) in {"t321": (3, 321), "t64": (4, 64), "t7": (2, 7)}.items():
            xs = lfcc_like(gen, B, 180, T)                     # stored layout [B,F,T]
            x = xs.transpose(1, 2)                              # what the harness feeds (predict.py:105)
            if cls is CNN2D:
                logits, emb = model(x, return_embedding=True)
                out[f"{tag}.embedding"] = emb.numpy()
            else:
                logits = model(x)
            out[f"{tag}.x_stored"] = xs.numpy()
            out[f"{tag}.logits"] = logits.numpy()
            lc = model(x.contiguous())
            assert torch.allclose(lc, logits, atol=1e-5)
        # per-layer activations, small T
        xs = lfcc_like(gen, 1, 180, 16)
        x = xs.transpose(1, 2)
        out["t16.x_stored"] = xs.numpy()
        if cls is CNN2D:
            h = x.unsqueeze(1)
            for i, layer in enumerate(model.conv):
                h = layer(h)
                if i in (4, 9, 12):                              # after pool/dropout 1, 2 and final ReLU
                    out[f"t16.a{(4, 9, 12).index(i) + 1}"] = h.numpy()
            out["t16.logits"] = model(x).numpy()
        else:
            h = x.transpose(1, 2)
            for i, layer in enumerate(model.conv):
                h = layer(h)
                if i in (3, 7, 10):
                    out[f"t16.h{(3, 7, 10).index(i) + 1}"] = h.numpy()
            out["t16.logits"] = model(x).numpy()
    print(name, "logit range", float(out["t321.logits"].min()), float(out["t321.logits"].max()))
    np.savez_compressed(os.path.join(OUT, name + "_eval.npz"), **out)


# ----------------------------------------------------------------------------------------- CAE eval
def make_cae_eval():
    gen = torch.Generator().manual_seed(4321)
    torch.manual_seed(1)
    model = ConvAutoencoder()
    randomise_bn(model, gen)
    warm_up(model, [torch.randn(2, 321, 180, generator=gen) for _ in range(2)])
    out = {"sd." + k: v for k, v in sd_np(model).items()}
    mse = torch.nn.MSELoss(reduction="none")
    with torch.no_grad():
        for tag, (B, T) in {"t321": (2, 321), "t64": (2, 64), "t70": (1, 70)}.items():
            x = torch.randn(B, T, 180, generator=gen) * 1.3 + 0.1
            recon, latent = model(x)
            out[f"{tag}.x"] = x.numpy()
            out[f"{tag}.recon"] = recon.numpy()
            out[f"{tag}.latent"] = latent.numpy()
            out[f"{tag}.mse"] = mse(recon, x).view(B, -1).mean(1).numpy()   # evaluation_cae.py:52-53
        # raw (un-normalised) input + normaliser: the fused z-score path
        raw = lfcc_like(gen, 2, 180, 321)
        norm = FeatureNormalizer().fit([r.transpose(0, 1) for r in raw])
        xz = torch.stack([norm.transform(r.transpose(0, 1)) for r in raw])
        recon, latent = model(xz)
        out["raw.x_stored"] = raw.numpy()
        out["raw.mean"] = norm.mean.numpy()
        out["raw.std"] = norm.std.numpy()
        out["raw.mse"] = mse(recon, xz).view(2, -1).mean(1).numpy()
        out["raw.recon"] = recon.numpy()
    np.savez_compressed(os.path.join(OUT, "cae_eval.npz"), **out)


# ----------------------------------------------------------------------------------------- train steps
def make_train(name, cls, B, T, loss_kind):
    gen = torch.Generator().manual_seed(99)
    out = {}
    for tag, eps in (("ls0", 0.0), ("ls05", 0.05)):
        torch.manual_seed(7)
        model = cls(in_features=180, dropout=0.0) if loss_kind == "bce" else cls()
        randomise_bn(model, torch.Generator().manual_seed(5))
        if loss_kind == "bce":
            with torch.no_grad():
                model.classifier.weight.mul_(40.0)
        model.train()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)   # train.py:326-328
        if tag == "ls0":
            out["init." + "sd"] = np.array(0)
            for k, v in sd_np(model).items():
                out["init.sd." + k] = v
        g2 = torch.Generator().manual_seed(2024)
        for step in (1, 2, 3):
            if loss_kind == "bce":
                xs = lfcc_like(g2, B, 180, T)
                x = xs.transpose(1, 2)
                y = (torch.rand(B, generator=g2) > 0.5).float()
                logits = model(x).squeeze(-1)
                ys = y * (1.0 - eps) + 0.5 * eps if eps > 0 else y            # train.py:311-315
                loss = torch.nn.BCEWithLogitsLoss()(logits, ys)
            else:
                x = torch.randn(B, T, 180, generator=g2)
                xs, y = x, torch.zeros(B)
                recon, _ = model(x)
                logits = recon
                loss = torch.nn.MSELoss()(recon, x)                             # train_cae.py:203
            opt.zero_grad()
            if step == 1 and loss_kind == "bce":
                logits.retain_grad()
            loss.backward()
            if step == 1:
                out[f"{tag}.x"] = xs.numpy()
                out[f"{tag}.y"] = y.numpy()
                out[f"{tag}.loss"] = np.float32(loss.item())
                if loss_kind == "bce":
                    out[f"{tag}.logits"] = logits.detach().numpy()
                    out[f"{tag}.dlogits"] = logits.grad.numpy()
                for k, p in model.named_parameters():
                    out[f"{tag}.grad.{k}"] = p.grad.numpy().copy()
            else:
                out[f"{tag}.x{step}"] = xs.numpy()
                out[f"{tag}.y{step}"] = y.numpy()
            opt.step()
            if step in (1, 3):
                for k, v in sd_np(model).items():
                    out[f"{tag}.after{step}.{k}"] = v
        if loss_kind == "mse":
            break                                                                # no label smoothing for the CAE
    np.savez_compressed(os.path.join(OUT, name + "_train.npz"), **out)


# ----------------------------------------------------------------------------------------- host-side pieces
def make_host():
    out = {}
    cases = {
        "sep": ([.1, .2, .8, .9], [0, 0, 1, 1]),
        "mix": ([.1, .4, .35, .8], [0, 0, 1, 1]),
        "inv": ([.9, .8, .2, .1], [0, 0, 1, 1]),
        "one": ([.3, .6], [1, 1]),
        "tie": ([.5] * 4, [0, 1, 0, 1]),
    }
    rng = np.random.default_rng(0)
    l = (rng.random(2000) > .55)
    s = rng.normal(2 * l, 1)
    cases["rng"] = (s.tolist(), l.astype(int).tolist())
    for k, (sc, lb) in cases.items():
        r1 = ref_eval_scripts.calculate_eer(sc, lb)
        r2 = ref_eval_src.calculate_eer(sc, lb)
        assert r1 == r2
        out[f"eer.{k}.scores"] = np.array(sc, np.float64)
        out[f"eer.{k}.labels"] = np.array(lb, np.int64)
        out[f"eer.{k}.result"] = np.array(r1, np.float64)
        thr = r1[1]
        out[f"eer.{k}.confusion"] = np.array(ref_eval_scripts.confusion_at_threshold(sc, lb, thr), np.float64)
    # normaliser
    g = torch.Generator().manual_seed(3)
    feats = [lfcc_like(g, 50 + 7 * i, 180) for i in range(5)]
    nz = FeatureNormalizer().fit(feats)
    out["norm.feats"] = torch.cat(feats, 0).numpy()
    out["norm.lens"] = np.array([f.shape[0] for f in feats])
    out["norm.mean"] = nz.mean.numpy()
    out["norm.std"] = nz.std.numpy()
    out["norm.t0"] = nz.transform(feats[0]).numpy()
    # fusion
    sup = rng.random(300)
    cae = rng.normal(0.4, 0.1, 300)
    lab = (rng.random(300) > 0.5).astype(int)
    sup = np.where(lab == 1, sup * 0.5 + 0.5, sup * 0.7)
    out["fuse.sup"], out["fuse.cae"], out["fuse.labels"] = sup, cae, lab
    out["fuse.sup_norm"] = normalise_scores(sup)
    out["fuse.cae_norm"] = normalise_scores(cae)
    out["fuse.const_norm"] = normalise_scores(np.full(5, 0.25))
    table = []
    for a in np.linspace(0.0, 1.0, 21):                                          # hybrid_ensemble.py:139-151
        e, _ = ref_eval_src.calculate_eer((a * out["fuse.sup_norm"] + (1 - a) * out["fuse.cae_norm"]).tolist(),
                                          lab.tolist())
        table.append((a, e))
    out["fuse.table"] = np.array(table)
    out["fuse.ens_mean"] = np.mean([sup, cae], axis=0)                           # ensemble.py:121
    np.savez_compressed(os.path.join(OUT, "host.npz"), **out)


if __name__ == "__main__":
    make_classifier_eval("cnn2d", CNN2D, 60.0)
    make_classifier_eval("cnn1d", CNN1D, 25.0)
    make_cae_eval()
    make_train("cnn2d", CNN2D, 4, 16, "bce")
    make_train("cnn1d", CNN1D, 4, 32, "bce")
    make_train("cae", ConvAutoencoder, 2, 32, "mse")
    make_host()
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")
