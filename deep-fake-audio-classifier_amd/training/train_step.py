"""Train-mode path of the HIP models.

Two ways to use it:
  * drop-in (reference semantics, src/train.py:71-76): `logits = model(x); loss = criterion(logits, y);
    optimizer.zero_grad(); loss.backward(); optimizer.step()` -- `model(x)` in train mode goes through
    `Cnn2dTrainFunction`, a torch.autograd.Function whose forward/backward are the C-ABI calls
    dfa_cnn2d_forward_train / dfa_cnn2d_backward; any torch criterion and optimizer work unchanged.
  * native (`NativeTrainer`): forward, BCE-with-smoothing, backward, ONE all-reduce of the flat gradient buffer
    (RCCL over xGMI when torch.distributed is initialised with backend "nccl") and the fused AdamW kernel,
    with no autograd graph and no per-parameter optimizer loop.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from .._params import tensors_signature


def _train_ws(model, ctx, nbytes):
    ws = getattr(model, "_train_ws", None)
    if ws is None or ws.numel() < nbytes or ws.device.index != ctx.index:
        model._train_ws = None
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=torch.device("cuda", ctx.index))
        model._train_ws = ws
    return ws


def _bind_cnn2d(model, ctx):
    """dfa_cnn2d_set_params with the CURRENT tensors (the kernels read weights and update running stats in place)."""
    ts = model._abi_tensors()
    for t in ts:
        if t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("CNN2D parameters must be contiguous float32 tensors on the GPU (model.to('cuda'))")
    sig = (ctx.index, tuple(t.data_ptr() for t in ts))
    stale = ctx.owner_changed("cnn2d", model)
    if getattr(model, "_bound", None) != sig or stale:
        arr = _lib.ptr_array([t.detach() for t in ts])
        _lib.check(ctx.handle, ctx.lib.dfa_cnn2d_set_params(ctx.handle, arr, len(ts), model.in_features,
                                                            model.base_channels))
        model._bound = sig
    model._prepared = None   # eval-mode folded images are stale after any training step


def _grad_sink(model):
    """FlatTrainer's gradient views when every parameter's .grad still IS its view of the flat buffer: the C-ABI backward then
    writes the gradients straight into the all-reduce payload (it writes, never accumulates) and autograd gets no tensors to
    add -- no per-parameter allocation, no add kernels, no memset.  None = plain autograd semantics (fresh tensors returned)."""
    sink = model.__dict__.get("_flat_grad_sink")
    if sink is None:
        return None
    for p, g in zip(model.parameters(), sink):
        if p.grad is None or p.grad.data_ptr() != g.data_ptr():
            return None
    return sink


def _next_dropout_offset(model, n_elems):
    off = getattr(model, "_drop_offset", 0)
    model._drop_offset = off + (n_elems + 3) // 4 + 1
    return off


def cnn2d_forward_train_raw(model, x, update_running_stats=True):
    """Run dfa_cnn2d_forward_train; returns (logits[B,1], ctx, workspace).  Stamps model._train_gen."""
    if x.device.type != "cuda":
        raise RuntimeError("dfa_amd.CNN2D runs on the GPU only: move the input with .to('cuda')")
    B, T, F = x.shape
    ctx = _lib.Context.get(x.device)
    with torch.cuda.device(ctx.index):
        ctx.use_current_stream()
        _bind_cnn2d(model, ctx)
        prec = _lib.PRECISIONS[model.precision]
        nbytes = ctx.lib.dfa_cnn2d_train_workspace_bytes(ctx.handle, B, T, F, prec)
        if nbytes == 0:
            raise ValueError(f"bad training shape (B={B}, T={T}, F={F})")
        ws = _train_ws(model, ctx, nbytes)
        logits = torch.empty((B, 1), dtype=torch.float32, device=x.device)
        seed = getattr(model, "_drop_seed", None)
        if seed is None:
            seed = model._drop_seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        offset = _next_dropout_offset(model, B * (T // 2) * F * 32)
        sb, st, sf = x.stride()
        code = ctx.lib.dfa_cnn2d_forward_train(
            ctx.handle, C.c_void_p(x.data_ptr()), _lib.x_dtype_code(x), B, T, F, sb, st, sf, prec,
            float(model.dropout), seed, offset, 0.1, int(update_running_stats), C.c_void_p(logits.data_ptr()), None,
            C.c_void_p(ws.data_ptr()), ws.numel())
        _lib.check(ctx.handle, code)
        model._train_gen = ctx.next_train_gen("cnn2d")
        model._train_shape = (B, T, F, prec, x.dtype)
        if update_running_stats:      # one multi-tensor launch instead of three
            torch._foreach_add_([model.conv[i].num_batches_tracked for i in model._BN_IDX], 1)
    return logits, ctx, ws


def cnn2d_backward_raw(model, x, dlogits, grad_tensors, ctx, ws, gen=None):
    B, T, F = x.shape
    ctx.check_train_gen("cnn2d", model._train_gen if gen is None else gen, model)
    if (B, T, F, _lib.PRECISIONS[model.precision], x.dtype) != getattr(model, "_train_shape", None):
        raise RuntimeError("backward called with a batch shape / precision / dtype other than its forward's")
    with torch.cuda.device(ctx.index):
        ctx.use_current_stream()
        arr = _lib.ptr_array(grad_tensors)
        sb, st, sf = x.stride()
        code = ctx.lib.dfa_cnn2d_backward(ctx.handle, C.c_void_p(x.data_ptr()), _lib.x_dtype_code(x), B, T, F, sb, st,
                                          sf, C.c_void_p(dlogits.data_ptr()), arr, len(grad_tensors),
                                          C.c_void_p(ws.data_ptr()), ws.numel())
        _lib.check(ctx.handle, code)


class Cnn2dTrainFunction(torch.autograd.Function):
    @staticmethod
    def forward(fctx, x, model, *params):
        logits, ctx, ws = cnn2d_forward_train_raw(model, x)
        fctx.model, fctx.x, fctx.ctx, fctx.ws, fctx.gen = model, x, ctx, ws, model._train_gen
        return logits

    @staticmethod
    def backward(fctx, dlogits):
        model = fctx.model
        grads = [torch.empty_like(p) for p in model.parameters()]
        cnn2d_backward_raw(model, fctx.x, dlogits.contiguous().float(), grads, fctx.ctx, fctx.ws, fctx.gen)
        return (None, None, *grads)


def cnn2d_train_forward(model, x, return_embedding=False):
    if return_embedding:
        raise NotImplementedError("return_embedding=True is an eval-mode feature (src/embedding_anomaly.py:61)")
    return Cnn2dTrainFunction.apply(x, model, *model.parameters())


def _bind_cnn1d(model, ctx):
    ts = model._abi_tensors()
    for t in ts:
        if t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("CNN1D parameters must be contiguous float32 tensors on the GPU (model.to('cuda'))")
    sig = (ctx.index, tuple(t.data_ptr() for t in ts))
    stale = ctx.owner_changed("cnn1d", model)
    if getattr(model, "_bound", None) != sig or stale:
        arr = _lib.ptr_array([t.detach() for t in ts])
        _lib.check(ctx.handle, ctx.lib.dfa_cnn1d_set_params(ctx.handle, arr, len(ts), model.in_features,
                                                            model.base_channels))
        model._bound = sig
    model._prepared = None


def cnn1d_forward_train_raw(model, x):
    """Run dfa_cnn1d_forward_train; returns (logits[B,1], ctx, workspace, generation)."""
    if x.device.type != "cuda":
        raise RuntimeError("dfa_amd.CNN1D runs on the GPU only: move the input with .to('cuda')")
    if x.dtype != torch.float32:
        raise ValueError(f"CNN1D takes float32 input, got {x.dtype}")
    B, T, F = x.shape
    ctx = _lib.Context.get(x.device)
    with torch.cuda.device(ctx.index):
        ctx.use_current_stream()
        _bind_cnn1d(model, ctx)
        nbytes = ctx.lib.dfa_cnn1d_train_workspace_bytes(ctx.handle, B, T, F)
        ws = _train_ws(model, ctx, nbytes)
        logits = torch.empty((B, 1), dtype=torch.float32, device=x.device)
        seed = getattr(model, "_drop_seed", None)
        if seed is None:
            seed = model._drop_seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        offset = _next_dropout_offset(model, B * 64 * T)
        sb, st, sf = x.stride()
        _lib.check(ctx.handle, ctx.lib.dfa_cnn1d_forward_train(
            ctx.handle, C.c_void_p(x.data_ptr()), _lib.DTYPE_F32, B, T, F, sb, st, sf, float(model.dropout), seed,
            offset, 0.1, 1, C.c_void_p(logits.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel()))
        torch._foreach_add_([model.conv[i].num_batches_tracked for i in model._BN_IDX], 1)
    return logits, ctx, ws, ctx.next_train_gen("cnn1d")


def cnn1d_backward_raw(model, x, dlogits, grad_tensors, ctx, ws, gen):
    """dfa_cnn1d_backward with the 14 gradients WRITTEN into grad_tensors (e.g. views of one flat buffer)."""
    ctx.check_train_gen("cnn1d", gen, model)
    B, T, F = x.shape
    with torch.cuda.device(ctx.index):
        ctx.use_current_stream()
        sb, st, sf = x.stride()
        _lib.check(ctx.handle, ctx.lib.dfa_cnn1d_backward(
            ctx.handle, C.c_void_p(x.data_ptr()), _lib.DTYPE_F32, B, T, F, sb, st, sf, C.c_void_p(dlogits.data_ptr()),
            _lib.ptr_array(grad_tensors), len(grad_tensors), C.c_void_p(ws.data_ptr()), ws.numel()))


class Cnn1dTrainFunction(torch.autograd.Function):
    @staticmethod
    def forward(fctx, x, model, *params):
        logits, ctx, ws, gen = cnn1d_forward_train_raw(model, x)
        fctx.model, fctx.x, fctx.ctx, fctx.ws, fctx.gen = model, x, ctx, ws, gen
        return logits

    @staticmethod
    def backward(fctx, dlogits):
        model = fctx.model
        sink = _grad_sink(model)
        grads = sink if sink is not None else [torch.empty_like(p) for p in model.parameters()]
        cnn1d_backward_raw(model, fctx.x, dlogits.contiguous().float(), grads, fctx.ctx, fctx.ws, fctx.gen)
        return (None, None, *([None] * len(grads) if sink is not None else grads))


def cnn1d_train_forward(model, x):
    return Cnn1dTrainFunction.apply(x, model, *model.parameters())


def _bind_cae(model, ctx):
    ts = model._abi_tensors()
    for t in ts:
        if t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("ConvAutoencoder parameters must be contiguous float32 tensors on the GPU")
    sig = (ctx.index, tuple(t.data_ptr() for t in ts))
    stale = ctx.owner_changed("cae", model)
    if getattr(model, "_bound", None) != sig or stale:
        arr = _lib.ptr_array([t.detach() for t in ts])
        _lib.check(ctx.handle, ctx.lib.dfa_cae_set_params(ctx.handle, arr, len(ts), model.base_channels))
        model._bound = sig
    model._prepared = None


def cae_forward_train_raw(model, x, want_recon=True, want_latent=True, want_mse=False):
    """Run dfa_cae_forward_train; returns (recon | None, latent | None, mse[B] | None, ctx, workspace, generation)."""
    if x.device.type != "cuda":
        raise RuntimeError("dfa_amd.ConvAutoencoder runs on the GPU only: move the input with .to('cuda')")
    B, T, F = x.shape
    ctx = _lib.Context.get(x.device)
    with torch.cuda.device(ctx.index):
        ctx.use_current_stream()
        _bind_cae(model, ctx)
        prec = _lib.PRECISIONS[model.precision]
        nbytes = ctx.lib.dfa_cae_train_workspace_bytes(ctx.handle, B, T, F, prec)
        if nbytes == 0:
            raise ValueError(f"bad auto-encoder training shape (B={B}, T={T}, F={F}): need T >= 16 and F = 16k+4")
        ws = _train_ws(model, ctx, nbytes)
        recon = torch.empty((B, T, F), dtype=torch.float32, device=x.device) if want_recon else None
        latent = torch.empty((B, 8 * model.base_channels, T // 16, F // 16), dtype=torch.float32, device=x.device) \
            if want_latent else None
        mse = torch.empty(B, dtype=torch.float32, device=x.device) if want_mse else None
        sb, st, sf = x.stride()

        def ptr(t):
            return C.c_void_p(t.data_ptr()) if t is not None else None
        _lib.check(ctx.handle, ctx.lib.dfa_cae_forward_train(
            ctx.handle, C.c_void_p(x.data_ptr()), _lib.x_dtype_code(x), B, T, F, sb, st, sf, prec, 0.1, 1,
            ptr(recon), ptr(latent), ptr(mse), C.c_void_p(ws.data_ptr()), ws.numel()))
        bns = [model.encoder[bi].num_batches_tracked for _, bi in model._ENC]
        bns += [model.decoder[bi].num_batches_tracked for _, bi in model._DEC if bi is not None]
        torch._foreach_add_(bns, 1)
    return recon, latent, mse, ctx, ws, ctx.next_train_gen("cae")


def cae_backward_raw(model, x, drecon, grad_tensors, ctx, ws, gen):
    """dfa_cae_backward with the 30 gradients WRITTEN into grad_tensors.  drecon = None: the loss is MSELoss(recon, x)
    (src/train_cae.py:67-68) and its gradient is formed inside the decoder's last backward kernel."""
    ctx.check_train_gen("cae", gen, model)
    B, T, F = x.shape
    with torch.cuda.device(ctx.index):
        ctx.use_current_stream()
        sb, st, sf = x.stride()
        _lib.check(ctx.handle, ctx.lib.dfa_cae_backward(
            ctx.handle, C.c_void_p(x.data_ptr()), _lib.x_dtype_code(x), B, T, F, sb, st, sf,
            C.c_void_p(drecon.data_ptr()) if drecon is not None else None, _lib.ptr_array(grad_tensors), len(grad_tensors),
            C.c_void_p(ws.data_ptr()), ws.numel()))


class CaeTrainFunction(torch.autograd.Function):
    """(reconstruction, latent) = ConvAutoencoder(x) in train mode; the gradient flows through the reconstruction
    (src/train_cae.py:67-71 uses MSELoss(recon, x)); the latent map is returned for inspection only."""

    @staticmethod
    def forward(fctx, x, model, *params):
        recon, latent, _, ctx, ws, gen = cae_forward_train_raw(model, x)
        fctx.model, fctx.x, fctx.ctx, fctx.ws, fctx.gen = model, x, ctx, ws, gen
        fctx.mark_non_differentiable(latent)
        return recon, latent

    @staticmethod
    def backward(fctx, drecon, _dlatent):
        model = fctx.model
        sink = _grad_sink(model)
        grads = sink if sink is not None else [torch.empty_like(p) for p in model.parameters()]
        cae_backward_raw(model, fctx.x, drecon.contiguous().float(), grads, fctx.ctx, fctx.ws, fctx.gen)
        return (None, None, *([None] * len(grads) if sink is not None else grads))


def cae_train_forward(model, x):
    return CaeTrainFunction.apply(x, model, *model.parameters())


class _FlatAdamW:
    """Parameters re-homed into ONE flat fp32 buffer (each nn.Parameter becomes a view), gradients in ONE flat buffer,
    AdamW moments alongside: the data-parallel exchange is a single all-reduce of `flat_g` and the update a single
    fused kernel (dfa_adamw_step) whatever the model (464,644 B CNN2D, 195,204 B CNN1D, 2,246,532 B auto-encoder)."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, process_group=None):
        self.model, self.lr, self.betas, self.eps, self.wd = model, lr, betas, eps, weight_decay
        self.pg = process_group
        params = list(model.parameters())
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError(f"{type(self).__name__} needs the model on the GPU")
        n = sum(p.numel() for p in params)
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.grad_views, off = [], 0
        for p in params:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view_as(p)
            self.grad_views.append(self.flat_g[off:off + k].view_as(p))
            off += k
        self.step_count = 0
        self._sched_opt = None

    @property
    def world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.pg) if (dist.is_available() and dist.is_initialized()) else 1

    def _exchange_and_update(self):
        """SUM all-reduce of the flat gradient (RCCL over xGMI under the "nccl" backend), then fused AdamW with the
        1/world scale folded in."""
        world = self.world
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.pg)
        self.step_count += 1
        ctx = _lib.Context.get(self.flat_p.device)
        with torch.cuda.device(ctx.index):
            ctx.use_current_stream()
            _lib.check(ctx.handle, ctx.lib.dfa_adamw_step(
                ctx.handle, C.c_void_p(self.flat_p.data_ptr()), C.c_void_p(self.flat_g.data_ptr()),
                C.c_void_p(self.exp_avg.data_ptr()), C.c_void_p(self.exp_avg_sq.data_ptr()), self.flat_p.numel(),
                float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd),
                self.step_count, 1.0 / world))
        self.model._prepared = None

    # ---- torch.optim.AdamW-compatible views (checkpoints stay interchangeable, src/training/checkpoint.py:42-71) ----
    def state_dict(self):
        state, off = {}, 0
        for i, p in enumerate(self.model.parameters()):
            k = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)),
                        "exp_avg": self.exp_avg[off:off + k].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + k].view_as(p).clone()}
            off += k
        group = {"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.wd, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(state)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        off = 0
        for i, p in enumerate(self.model.parameters()):
            k = p.numel()
            st = sd["state"].get(i)
            if st is not None:
                self.exp_avg[off:off + k].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
                self.step_count = int(st["step"])
            off += k
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps, self.wd = g["lr"], tuple(g["betas"]), g["eps"], g["weight_decay"]
        if self._sched_opt is not None:       # the plateau scheduler's stand-in must resume from the restored lr too
            self._sched_opt.param_groups[0]["lr"] = float(self.lr)

    def plateau_scheduler(self, **kw):
        """torch's ReduceLROnPlateau driving THIS trainer's lr (src/train.py:332-341,520-525): the scheduler owns a
        one-group stand-in optimiser whose lr is copied back after every scheduler.step(metric)."""
        trainer = self
        self._sched_opt = torch.optim.SGD([torch.zeros(1, requires_grad=True)], lr=float(self.lr))

        class _Plateau(torch.optim.lr_scheduler.ReduceLROnPlateau):
            def step(self, metrics, *a, **k):
                out = super().step(metrics, *a, **k)
                trainer.lr = float(trainer._sched_opt.param_groups[0]["lr"])
                return out
        return _Plateau(self._sched_opt, **kw)


class FlatTrainer(_FlatAdamW):
    """Optimizer-shaped data-parallel engine for ANY dfa_amd model used through the autograd bridge (CNN1D with BCE,
    the auto-encoder with MSELoss, src/train_cae.py:58-82): `zero_grad(); loss.backward(); step()`.  Every parameter's
    .grad is a view of the flat gradient buffer, so autograd accumulates straight into the all-reduce payload."""

    def __init__(self, model, **kw):
        super().__init__(model, **kw)
        for p, g in zip(model.parameters(), self.grad_views):
            p.grad = g
        model.__dict__["_flat_grad_sink"] = self.grad_views     # CNN1D / auto-encoder bridges write into the views directly

    def zero_grad(self, set_to_none: bool = False):
        """Re-attach the views; no memset: the dfa_*_backward calls WRITE every gradient (one backward per step on this path)."""
        for p, g in zip(self.model.parameters(), self.grad_views):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def step(self):
        self._exchange_and_update()


class NativeTrainer(_FlatAdamW):
    """Whole classifier training step on the C ABI: forward_train -> BCE(smoothed) -> backward (gradients written straight
    into the views of the flat buffer) -> all-reduce -> fused AdamW, with no autograd graph (src/train.py:71-76).
    CNN2D and CNN1D (chosen by the model's class)."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, label_smoothing=0.0,
                 process_group=None, sync_bn=False):
        if not (0.0 <= label_smoothing < 0.5):
            raise ValueError("--label-smoothing must be in [0, 0.5)")          # src/train.py:308-309
        super().__init__(model, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, process_group=process_group)
        self.label_smoothing = label_smoothing
        # sync_bn (world > 1): BatchNorm statistics over the GLOBAL batch (two 2C-float all-reduces per layer and step through the
        # C ABI's hook): N ranks x B then train like one rank x N*B; the default is DistributedDataParallel's local statistics
        self.sync_bn = bool(sync_bn)
        _lib.Context.get(self.flat_p.device).set_bn_sync(process_group, enable=self.sync_bn)
        self.loss_buf = torch.zeros(1, dtype=torch.float32, device=self.flat_p.device)
        self.dlogits = None
        self.kind = "cnn1d" if type(model).__name__ == "CNN1D" else "cnn2d"

    def step(self, x, y):
        """One optimisation step on batch (x[B,T,F], y[B]); returns the (device) loss scalar of this rank's batch."""
        model = self.model
        model.train()
        if self.kind == "cnn1d":
            logits, ctx, ws, gen = cnn1d_forward_train_raw(model, x)
        else:
            logits, ctx, ws = cnn2d_forward_train_raw(model, x)
        B = x.shape[0]
        if self.dlogits is None or self.dlogits.numel() != B:
            self.dlogits = torch.empty(B, dtype=torch.float32, device=x.device)
        y = y.to(device=x.device, dtype=torch.float32).contiguous()
        with torch.cuda.device(ctx.index):
            _lib.check(ctx.handle, ctx.lib.dfa_bce_smooth_fwd_bwd(
                ctx.handle, C.c_void_p(logits.data_ptr()), C.c_void_p(y.data_ptr()), float(self.label_smoothing), B,
                C.c_void_p(self.loss_buf.data_ptr()), C.c_void_p(self.dlogits.data_ptr())))
        if self.kind == "cnn1d":
            cnn1d_backward_raw(model, x, self.dlogits, self.grad_views, ctx, ws, gen)
        else:
            cnn2d_backward_raw(model, x, self.dlogits, self.grad_views, ctx, ws)
        self._exchange_and_update()
        return self.loss_buf


class CaeNativeTrainer(_FlatAdamW):
    """Whole auto-encoder training step on the C ABI (src/train_cae.py:58-82: recon = model(x); MSELoss(recon, x); backward;
    AdamW): forward_train writes only the per-sample MSE, the backward forms 2 (recon - x) / N inside its first kernel
    (dfa_cae_backward with drecon = NULL) and writes the 30 gradients straight into the flat buffer, then ONE 2,246,532-byte
    all-reduce and the fused AdamW.  No reconstruction, no loss gradient, no autograd graph and no torch elementwise kernel."""

    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, process_group=None, sync_bn=False):
        super().__init__(model, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, process_group=process_group)
        self.sync_bn = bool(sync_bn)      # BatchNorm statistics over the global batch (NativeTrainer's docstring)
        _lib.Context.get(self.flat_p.device).set_bn_sync(process_group, enable=self.sync_bn)

    def step(self, x):
        """One optimisation step on the (z-scored) batch x[B,T,F]; returns the device scalar MSELoss(recon, x) of this rank."""
        model = self.model
        model.train()
        _, _, mse, ctx, ws, gen = cae_forward_train_raw(model, x, want_recon=False, want_latent=False, want_mse=True)
        cae_backward_raw(model, x, None, self.grad_views, ctx, ws, gen)
        self._exchange_and_update()
        return mse.mean()          # every sample has T*F elements: the mean of the per-sample MSEs is nn.MSELoss's mean


def make_cae_trainer(model, **kw):
    return CaeNativeTrainer(model, **kw)
