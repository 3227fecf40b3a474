"""CNN1D -- MI355X counterpart of the reference's src/model_cnn1d.py:5-46.

Same constructor / state_dict / call contract (`model(x[B,T,F]) -> logits[B,1]`).  The reference transposes to
[B,F,T] for Conv1d (model_cnn1d.py:40); here that transpose is free: the kernels address x through its strides, and
the stored feature layout [B,180,321] already is channel-major.  Arithmetic is fp32 (this path is bound by reading
the input once: 30.8 MFLOP per 231 KB utterance)."""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib
from ._params import BatchNormParams, ConvParams, LinearParams, Slots, tensors_signature


class CNN1D(nn.Module):
    _CONV_IDX = (0, 4, 8)     # reference nn.Sequential indices that own parameters (src/model_cnn1d.py:15-32)
    _BN_IDX = (1, 5, 9)

    def __init__(self, in_features=180, base_channels=32, num_classes=1, dropout=0.2):
        super().__init__()
        if num_classes != 1:
            raise ValueError("dfa_amd.CNN1D implements the binary head (num_classes=1) the reference trains")
        bc = base_channels
        chans = [(in_features, bc), (bc, 2 * bc), (2 * bc, 4 * bc)]
        slots = {}
        for ci, bi, (cin, cout) in zip(self._CONV_IDX, self._BN_IDX, chans):
            slots[ci] = ConvParams(cin, cout, (3,))
            slots[bi] = BatchNormParams(cout)
        self.conv = Slots(slots)
        self.classifier = LinearParams(4 * bc, num_classes)
        self.in_features, self.base_channels, self.dropout = in_features, base_channels, float(dropout)
        self._prepared = None

    def _abi_tensors(self):
        out = []
        for ci, bi in zip(self._CONV_IDX, self._BN_IDX):
            c, b = self.conv[ci], self.conv[bi]
            out += [c.weight, c.bias, b.weight, b.bias, b.running_mean, b.running_var]
        return out + [self.classifier.weight, self.classifier.bias]

    def _ensure_prepared(self, ctx):
        ts = self._abi_tensors()
        for t in ts:
            if t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("CNN1D parameters must be contiguous float32 tensors on the GPU "
                                   "(call model.to('cuda')); dfa_amd has no CPU path")
        sig = (ctx.index, tensors_signature(ts))
        stale = ctx.owner_changed("cnn1d", self)      # another model of this class used the ctx's weight slot
        if sig == self._prepared and not stale:
            return
        arr = _lib.ptr_array([t.detach() for t in ts])
        _lib.check(ctx.handle, ctx.lib.dfa_cnn1d_set_params(ctx.handle, arr, len(ts), self.in_features,
                                                            self.base_channels))
        _lib.check(ctx.handle, ctx.lib.dfa_cnn1d_prepare(ctx.handle))
        self._prepared = sig

    def forward(self, x):
        if x.dim() != 3:
            raise ValueError(f"CNN1D expects x of shape (B, T, F), got {tuple(x.shape)}")
        if self.training:
            from .training import cnn1d_train_forward
            return cnn1d_train_forward(self, x)
        if x.device.type != "cuda":
            raise RuntimeError("dfa_amd.CNN1D runs on the GPU only: move the input with .to('cuda')")
        if x.dtype != torch.float32:
            raise ValueError(f"CNN1D takes float32 input, got {x.dtype}")
        B, T, F = x.shape
        ctx = _lib.Context.get(x.device)
        with torch.cuda.device(ctx.index):
            ctx.use_current_stream()
            self._ensure_prepared(ctx)
            nbytes = ctx.lib.dfa_workspace_bytes(ctx.handle, _lib.MODEL_CNN1D, B, T, F, _lib.PREC_F32)
            ws = ctx.workspace(nbytes)
            logits = torch.empty((B, 1), dtype=torch.float32, device=x.device)
            sb, st, sf = x.stride()
            code = ctx.lib.dfa_cnn1d_forward(ctx.handle, C.c_void_p(x.data_ptr()), _lib.DTYPE_F32, B, T, F, sb, st, sf,
                                             C.c_void_p(logits.data_ptr()), C.c_void_p(ws.data_ptr()), ws.numel())
            _lib.check(ctx.handle, code)
        return logits


if __name__ == "__main__":
    model = CNN1D().to("cuda").eval()
    print(f"CNN1D output shape: {model(torch.randn(4, 321, 180, device='cuda')).shape}")
