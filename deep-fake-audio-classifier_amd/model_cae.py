"""ConvAutoencoder -- MI355X counterpart of the reference's src/model_cae.py:20-125.

Same constructor, state_dict keys (encoder.{0,1,4,5,8,9,12,13}.*, decoder.{0,1,3,4,6,7,9}.*; ConvTranspose2d weights
in torch's (Cin, Cout, 2, 2) layout) and call contract: `model(x[B,T,F]) -> (reconstruction[B,T,F], latent[B,256,T/16,F/16])`.

Beyond the reference: `model.score(x, mean=None, std=None)` returns the per-sample reconstruction MSE
(src/evaluation_cae.py:52-53) computed on the GPU with the FeatureNormalizer z-score fused into the loads, so the
anomaly scores of a raw feature batch need no normalised copy, no reconstruction tensor and no latent export.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
from torch import nn

from . import _lib
from ._params import BatchNormParams, ConvParams, Slots, tensors_signature


class ConvAutoencoder(nn.Module):
    _ENC = ((0, 1), (4, 5), (8, 9), (12, 13))     # (conv idx, bn idx) in the reference nn.Sequential (model_cae.py:32-56)
    _DEC = ((0, 1), (3, 4), (6, 7), (9, None))    # ConvTranspose2d idx, bn idx (model_cae.py:61-80)

    def __init__(self, base_channels: int = 32, precision=None):
        super().__init__()
        bc = base_channels
        enc_ch = [(1, bc), (bc, 2 * bc), (2 * bc, 4 * bc), (4 * bc, 8 * bc)]
        dec_ch = [(8 * bc, 4 * bc), (4 * bc, 2 * bc), (2 * bc, bc), (bc, 1)]
        enc, dec = {}, {}
        for (ci, bi), (cin, cout) in zip(self._ENC, enc_ch):
            enc[ci] = ConvParams(cin, cout, (3, 3))
            enc[bi] = BatchNormParams(cout)
        self.encoder = Slots(enc)
        for (ci, bi), (cin, cout) in zip(self._DEC, dec_ch):
            dec[ci] = ConvParams(cin, cout, (2, 2), transposed=True)
            if bi is not None:
                dec[bi] = BatchNormParams(cout)
        self.decoder = Slots(dec)
        self.base_channels = base_channels
        self.precision = (precision or os.environ.get("DFA_PRECISION", "fp32")).lower()
        if self.precision not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {self.precision!r}")
        self._prepared = None

    def set_precision(self, precision: str) -> "ConvAutoencoder":
        if precision.lower() not in _lib.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_lib.PRECISIONS)}, got {precision!r}")
        self.precision = precision.lower()
        return self

    def _abi_tensors(self):
        out = []
        for ci, bi in self._ENC:
            c, b = self.encoder[ci], self.encoder[bi]
            out += [c.weight, c.bias, b.weight, b.bias, b.running_mean, b.running_var]
        for ci, bi in self._DEC:
            c = self.decoder[ci]
            out += [c.weight, c.bias]
            if bi is not None:
                b = self.decoder[bi]
                out += [b.weight, b.bias, b.running_mean, b.running_var]
        return out

    def _ensure_prepared(self, ctx):
        ts = self._abi_tensors()
        for t in ts:
            if t.device.type != "cuda" or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("ConvAutoencoder parameters must be contiguous float32 tensors on the GPU "
                                   "(call model.to('cuda')); dfa_amd has no CPU path")
        sig = (ctx.index, self.precision, tensors_signature(ts))
        stale = ctx.owner_changed("cae", self)      # another model of this class used the ctx's weight slot
        if sig == self._prepared and not stale:
            return
        arr = _lib.ptr_array([t.detach() for t in ts])
        _lib.check(ctx.handle, ctx.lib.dfa_cae_set_params(ctx.handle, arr, len(ts), self.base_channels))
        _lib.check(ctx.handle, ctx.lib.dfa_cae_prepare(ctx.handle, _lib.PRECISIONS[self.precision]))
        self._prepared = sig

    def _run(self, x, mean, std, want_recon, want_latent, want_mse):
        if x.dim() != 3:
            raise ValueError(f"ConvAutoencoder expects x of shape (B, T, F), got {tuple(x.shape)}")
        if x.device.type != "cuda":
            raise RuntimeError("dfa_amd.ConvAutoencoder runs on the GPU only: move the input with .to('cuda')")
        B, T, F = x.shape
        ctx = _lib.Context.get(x.device)
        with torch.cuda.device(ctx.index):
            ctx.use_current_stream()
            self._ensure_prepared(ctx)
            prec = _lib.PRECISIONS[self.precision]
            nbytes = ctx.lib.dfa_workspace_bytes(ctx.handle, _lib.MODEL_CAE, B, T, F, prec)
            ws = ctx.workspace(max(nbytes, 256))
            dev = x.device
            recon = torch.empty((B, T, F), dtype=torch.float32, device=dev) if want_recon else None
            latent = torch.empty((B, 8 * self.base_channels, T // 16, F // 16), dtype=torch.float32, device=dev) \
                if want_latent else None
            mse = torch.empty((B,), dtype=torch.float32, device=dev) if want_mse else None
            if mean is not None:
                mean = mean.to(device=dev, dtype=torch.float32).contiguous()
                std = std.to(device=dev, dtype=torch.float32).contiguous()
                if mean.numel() != F or std.numel() != F:
                    raise ValueError(f"normaliser statistics must have {F} entries")

            def ptr(t):
                return C.c_void_p(t.data_ptr() if t is not None else None)
            sb, st, sf = x.stride()
            code = ctx.lib.dfa_cae_forward(ctx.handle, ptr(x), _lib.x_dtype_code(x), B, T, F, sb, st, sf, ptr(mean),
                                           ptr(std), ptr(recon), ptr(latent), ptr(mse), ptr(ws), ws.numel())
            _lib.check(ctx.handle, code)
        return recon, latent, mse

    def forward(self, x: torch.Tensor):
        """x: (B, T, F) normalised spectrogram -> (reconstruction (B, T, F), latent (B, 256, T/16, F/16))."""
        if self.training:
            from .training import cae_train_forward
            return cae_train_forward(self, x)
        recon, latent, _ = self._run(x, None, None, True, True, False)
        return recon, latent

    @torch.no_grad()
    def score(self, x: torch.Tensor, mean: torch.Tensor | None = None, std: torch.Tensor | None = None):
        """Per-sample reconstruction MSE [B].  With mean/std the input is the RAW feature view and the z-score is fused."""
        if (mean is None) != (std is None):
            raise ValueError("mean and std must be given together")
        return self._run(x, mean, std, False, False, True)[2]


if __name__ == "__main__":
    model = ConvAutoencoder().to("cuda").eval()
    x = torch.randn(4, 321, 180, device="cuda")
    recon, latent = model(x)
    print(f"Latent shape: {latent.shape}  Reconstruction shape: {recon.shape}")
    assert x.shape == recon.shape
    print(f"Total parameters: {sum(p.numel() for p in model.parameters()):,}")
