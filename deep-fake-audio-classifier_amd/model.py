"""CNN2D -- MI355X counterpart of the reference's src/model.py:5-42.

Same constructor, same state_dict keys/shapes/parameter order, same call contract
(`model(x[B,T,F]) -> logits[B,1]`, `model(x, return_embedding=True) -> (logits, emb[B, 128*F])`), but `forward`
enqueues the hand-written HIP kernels of libdfa_hip.so (conv1 -> MFMA block 2 -> MFMA block 3 + time mean ->
linear) instead of torch layers.  Input may be the non-contiguous transposed view the reference harness feeds
(src/predict.py:105): strides are passed through, nothing is copied.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
from torch import nn

from . import _lib
from ._params import BatchNormParams, ConvParams, LinearParams, Slots, tensors_signature


class CNN2D(nn.Module):
    """2-D CNN over the time x feature grid (reference: src/model.py:5-42).

    precision: "fp32" (default; exact-fp32 MFMA, logits within 1e-4 of the reference), "bf16" (bf16 storage, fp32
    accumulate: the throughput mode of BASELINE.json configs[1]) or "bf16x3" (eval only: hi + lo bf16 operands, three bf16
    MFMAs per product -- logits within 1e-4 of the reference at several times the fp32 rate).  Env DFA_PRECISION overrides
    the default.
    """

    # reference nn.Sequential indices that own parameters (src/model.py:13-29)
    _CONV_IDX = (0, 5, 10)
    _BN_IDX = (1, 6, 11)

    def __init__(self, in_features=180, base_channels=32, num_classes=1, dropout=0.2, precision=None):
        super().__init__()
        if num_classes != 1:
            raise ValueError("dfa_amd.CNN2D implements the binary head (num_classes=1) the reference trains")
        bc = base_channels
        chans = [(1, bc), (bc, 2 * bc), (2 * bc, 4 * bc)]
        slots = {}
        for ci, bi, (cin, cout) in zip(self._CONV_IDX, self._BN_IDX, chans):
            slots[ci] = ConvParams(cin, cout, (3, 3))      # draws RNG in reference order: conv weight, conv bias
            slots[bi] = BatchNormParams(cout)
        self.conv = Slots(slots)
        self.classifier = LinearParams(4 * bc * in_features, num_classes)
        self.in_features = in_features
        self.base_channels = base_channels
        self.dropout = float(dropout)
        self.precision = (precision or os.environ.get("DFA_PRECISION", "fp32")).lower()
        if self.precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {self.precision!r}")
        self._prepared = None   # (device index, precision, tensors signature)

    # ---- parameter plumbing ------------------------------------------------------------------------------------
    def _abi_tensors(self):
        """The 20 tensors of dfa_cnn2d_set_params, in include/dfa_hip.h order."""
        out = []
        for ci, bi in zip(self._CONV_IDX, self._BN_IDX):
            c, b = self.conv[ci], self.conv[bi]
            out += [c.weight, c.bias, b.weight, b.bias, b.running_mean, b.running_var]
        out += [self.classifier.weight, self.classifier.bias]
        return out

    def set_precision(self, precision: str) -> "CNN2D":
        precision = precision.lower()
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {precision!r}")
        self.precision = precision
        return self

    def _ensure_prepared(self, ctx: "_lib.Context"):
        ts = self._abi_tensors()
        for t in ts:
            if t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("CNN2D parameters must be contiguous float32 tensors on the GPU "
                                   "(call model.to('cuda')); dfa_amd has no CPU path")
        sig = (ctx.index, self.precision, tensors_signature(ts))
        stale = ctx.owner_changed("cnn2d", self)      # another model of this class used the ctx's weight slot
        if sig == self._prepared and not stale:
            return
        arr = _lib.ptr_array([t.detach() for t in ts])
        _lib.check(ctx.handle, ctx.lib.dfa_cnn2d_set_params(ctx.handle, arr, len(ts), self.in_features,
                                                            self.base_channels))
        _lib.check(ctx.handle, ctx.lib.dfa_cnn2d_prepare(ctx.handle, _lib.PRECISIONS[self.precision]))
        self._prepared = sig

    # ---- forward -----------------------------------------------------------------------------------------------
    def forward(self, x, return_embedding=False):
        if x.dim() != 3:
            raise ValueError(f"CNN2D expects x of shape (B, T, F), got {tuple(x.shape)}")
        if self.training:
            from .training import cnn2d_train_forward  # train-mode path (batch-stat BN, dropout, autograd)
            return cnn2d_train_forward(self, x, return_embedding)
        return self._eval_forward(x, return_embedding)

    def _eval_forward(self, x, return_embedding):
        if x.device.type != "cuda":
            raise RuntimeError("dfa_amd.CNN2D runs on the GPU only: move the input with .to('cuda')")
        B, T, F = x.shape
        ctx = _lib.Context.get(x.device)
        with torch.cuda.device(ctx.index):
            ctx.use_current_stream()
            self._ensure_prepared(ctx)
            prec = _lib.PRECISIONS[self.precision]
            nbytes = ctx.lib.dfa_workspace_bytes(ctx.handle, _lib.MODEL_CNN2D, B, T, F, prec)
            ws = ctx.workspace(nbytes)
            logits = torch.empty((B, 1), dtype=torch.float32, device=x.device)
            emb = torch.empty((B, 4 * self.base_channels * F), dtype=torch.float32, device=x.device) \
                if return_embedding else None
            sb, st, sf = x.stride()
            code = ctx.lib.dfa_cnn2d_forward(
                ctx.handle, C.c_void_p(x.data_ptr()), _lib.x_dtype_code(x), B, T, F, sb, st, sf,
                C.c_void_p(logits.data_ptr()), C.c_void_p(emb.data_ptr() if emb is not None else None),
                C.c_void_p(ws.data_ptr()), ws.numel())
            _lib.check(ctx.handle, code)
        if return_embedding:
            return logits, emb
        return logits


if __name__ == "__main__":
    model = CNN2D().to("cuda").eval()
    x = torch.randn(4, 321, 180, device="cuda")
    print(f"CNN2D output shape: {model(x).shape}")
