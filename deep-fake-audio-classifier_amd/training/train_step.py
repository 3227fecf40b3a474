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
    if getattr(model, "_bound", None) != sig:
        arr = _lib.ptr_array([t.detach() for t in ts])
        _lib.check(ctx.handle, ctx.lib.dfa_cnn2d_set_params(ctx.handle, arr, len(ts), model.in_features,
                                                            model.base_channels))
        model._bound = sig
    model._prepared = None   # eval-mode folded images are stale after any training step


def _next_dropout_offset(model, n_elems):
    off = getattr(model, "_drop_offset", 0)
    model._drop_offset = off + (n_elems + 3) // 4 + 1
    return off


def cnn2d_forward_train_raw(model, x, update_running_stats=True):
    """Run dfa_cnn2d_forward_train; returns (logits[B,1], ctx, workspace)."""
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
        if update_running_stats:
            for i in model._BN_IDX:
                model.conv[i].num_batches_tracked += 1
    return logits, ctx, ws


def cnn2d_backward_raw(model, x, dlogits, grad_tensors, ctx, ws):
    B, T, F = x.shape
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
        fctx.model, fctx.x, fctx.ctx, fctx.ws = model, x, ctx, ws
        return logits

    @staticmethod
    def backward(fctx, dlogits):
        model = fctx.model
        grads = [torch.empty_like(p) for p in model.parameters()]
        cnn2d_backward_raw(model, fctx.x, dlogits.contiguous().float(), grads, fctx.ctx, fctx.ws)
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
    if getattr(model, "_bound", None) != sig:
        arr = _lib.ptr_array([t.detach() for t in ts])
        _lib.check(ctx.handle, ctx.lib.dfa_cnn1d_set_params(ctx.handle, arr, len(ts), model.in_features,
                                                            model.base_channels))
        model._bound = sig
    model._prepared = None


class Cnn1dTrainFunction(torch.autograd.Function):
    @staticmethod
    def forward(fctx, x, model, *params):
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
            for i in model._BN_IDX:
                model.conv[i].num_batches_tracked += 1
        fctx.model, fctx.x, fctx.ctx, fctx.ws = model, x, ctx, ws
        return logits

    @staticmethod
    def backward(fctx, dlogits):
        model, x, ctx, ws = fctx.model, fctx.x, fctx.ctx, fctx.ws
        grads = [torch.empty_like(p) for p in model.parameters()]
        B, T, F = x.shape
        d = dlogits.contiguous().float()
        with torch.cuda.device(ctx.index):
            ctx.use_current_stream()
            sb, st, sf = x.stride()
            _lib.check(ctx.handle, ctx.lib.dfa_cnn1d_backward(
                ctx.handle, C.c_void_p(x.data_ptr()), _lib.DTYPE_F32, B, T, F, sb, st, sf, C.c_void_p(d.data_ptr()),
                _lib.ptr_array(grads), len(grads), C.c_void_p(ws.data_ptr()), ws.numel()))
        return (None, None, *grads)


def cnn1d_train_forward(model, x):
    return Cnn1dTrainFunction.apply(x, model, *model.parameters())


def _bind_cae(model, ctx):
    ts = model._abi_tensors()
    for t in ts:
        if t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("ConvAutoencoder parameters must be contiguous float32 tensors on the GPU")
    sig = (ctx.index, tuple(t.data_ptr() for t in ts))
    if getattr(model, "_bound", None) != sig:
        arr = _lib.ptr_array([t.detach() for t in ts])
        _lib.check(ctx.handle, ctx.lib.dfa_cae_set_params(ctx.handle, arr, len(ts), model.base_channels))
        model._bound = sig
    model._prepared = None


class CaeTrainFunction(torch.autograd.Function):
    """(reconstruction, latent) = ConvAutoencoder(x) in train mode; the gradient flows through the reconstruction
    (src/train_cae.py:67-71 uses MSELoss(recon, x)); the latent map is returned for inspection only."""

    @staticmethod
    def forward(fctx, x, model, *params):
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
            recon = torch.empty((B, T, F), dtype=torch.float32, device=x.device)
            latent = torch.empty((B, 8 * model.base_channels, T // 16, F // 16), dtype=torch.float32, device=x.device)
            sb, st, sf = x.stride()
            _lib.check(ctx.handle, ctx.lib.dfa_cae_forward_train(
                ctx.handle, C.c_void_p(x.data_ptr()), _lib.x_dtype_code(x), B, T, F, sb, st, sf, prec, 0.1, 1,
                C.c_void_p(recon.data_ptr()), C.c_void_p(latent.data_ptr()), None, C.c_void_p(ws.data_ptr()),
                ws.numel()))
            for _, bi in model._ENC:
                model.encoder[bi].num_batches_tracked += 1
            for _, bi in model._DEC:
                if bi is not None:
                    model.decoder[bi].num_batches_tracked += 1
        fctx.model, fctx.x, fctx.ctx, fctx.ws = model, x, ctx, ws
        fctx.mark_non_differentiable(latent)
        return recon, latent

    @staticmethod
    def backward(fctx, drecon, _dlatent):
        model, x, ctx, ws = fctx.model, fctx.x, fctx.ctx, fctx.ws
        grads = [torch.empty_like(p) for p in model.parameters()]
        B, T, F = x.shape
        d = drecon.contiguous().float()
        with torch.cuda.device(ctx.index):
            ctx.use_current_stream()
            sb, st, sf = x.stride()
            _lib.check(ctx.handle, ctx.lib.dfa_cae_backward(
                ctx.handle, C.c_void_p(x.data_ptr()), _lib.x_dtype_code(x), B, T, F, sb, st, sf,
                C.c_void_p(d.data_ptr()), _lib.ptr_array(grads), len(grads), C.c_void_p(ws.data_ptr()), ws.numel()))
        return (None, None, *grads)


def cae_train_forward(model, x):
    return CaeTrainFunction.apply(x, model, *model.parameters())


class NativeTrainer:
    """Whole CNN2D training step on the C ABI: forward_train -> BCE(smoothed) -> backward -> all-reduce -> fused AdamW.

    Parameters are re-homed into ONE flat fp32 buffer (each nn.Parameter becomes a view), gradients are produced into
    ONE flat buffer, so data-parallel training needs a single all-reduce of 464,644 bytes per step."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, label_smoothing=0.0,
                 process_group=None):
        if not (0.0 <= label_smoothing < 0.5):
            raise ValueError("--label-smoothing must be in [0, 0.5)")          # src/train.py:308-309
        self.model, self.lr, self.betas, self.eps, self.wd = model, lr, betas, eps, weight_decay
        self.label_smoothing = label_smoothing
        self.pg = process_group
        params = list(model.parameters())
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("NativeTrainer needs the model on the GPU")
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
        self.loss_buf = torch.zeros(1, dtype=torch.float32, device=dev)
        self.dlogits = None

    @property
    def world(self):
        import torch.distributed as dist
        return dist.get_world_size(self.pg) if (dist.is_available() and dist.is_initialized()) else 1

    def step(self, x, y):
        """One optimisation step on batch (x[B,T,F], y[B]); returns the (device) loss scalar of this rank's batch."""
        model = self.model
        model.train()
        logits, ctx, ws = cnn2d_forward_train_raw(model, x)
        B = x.shape[0]
        if self.dlogits is None or self.dlogits.numel() != B:
            self.dlogits = torch.empty(B, dtype=torch.float32, device=x.device)
        y = y.to(device=x.device, dtype=torch.float32).contiguous()
        with torch.cuda.device(ctx.index):
            _lib.check(ctx.handle, ctx.lib.dfa_bce_smooth_fwd_bwd(
                ctx.handle, C.c_void_p(logits.data_ptr()), C.c_void_p(y.data_ptr()), float(self.label_smoothing), B,
                C.c_void_p(self.loss_buf.data_ptr()), C.c_void_p(self.dlogits.data_ptr())))
        cnn2d_backward_raw(model, x, self.dlogits, self.grad_views, ctx, ws)
        world = self.world
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.pg)     # RCCL on ROCm ("nccl" backend)
        self.step_count += 1
        with torch.cuda.device(ctx.index):
            ctx.use_current_stream()
            _lib.check(ctx.handle, ctx.lib.dfa_adamw_step(
                ctx.handle, C.c_void_p(self.flat_p.data_ptr()), C.c_void_p(self.flat_g.data_ptr()),
                C.c_void_p(self.exp_avg.data_ptr()), C.c_void_p(self.exp_avg_sq.data_ptr()), self.flat_p.numel(),
                float(self.lr), float(self.betas[0]), float(self.betas[1]), float(self.eps), float(self.wd),
                self.step_count, 1.0 / world))
        model._prepared = None
        return self.loss_buf
