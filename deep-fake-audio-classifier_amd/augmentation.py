"""Train-time feature augmentations on [B, T, F] batches -- counterparts of src/augmentation.py:5-186.

Semantics kept: one random draw per BATCH (not per sample) from Python's `random` for the shift / mask spans and from
the torch generator for the channel mask and the jitter noise, so seeded runs pick the same spans as the reference.
They are cheap, memory-bound device ops that run before the first HIP kernel reads the batch.
"""
from __future__ import annotations

import random

import torch


def time_shift(features: torch.Tensor, max_shift_ratio: float = 0.1) -> torch.Tensor:
    """Circular shift along T by a random amount in [-max_shift_ratio*T, +max_shift_ratio*T]."""
    if max_shift_ratio <= 0 or features.shape[1] <= 1:
        return features
    limit = int(features.shape[1] * max_shift_ratio)
    if limit < 1:
        return features
    shift = random.randint(-limit, limit)
    return features if shift == 0 else torch.roll(features, shifts=shift, dims=1)


def channel_drop(features: torch.Tensor, drop_prob: float = 0.1) -> torch.Tensor:
    """Zero whole feature dims with probability drop_prob (one [1,1,F] mask for the batch)."""
    if drop_prob <= 0:
        return features
    keep = (torch.rand((1, 1, features.shape[2]), device=features.device) >= drop_prob).to(features.dtype)
    return features * keep


def gaussian_jitter(features: torch.Tensor, std: float = 0.01) -> torch.Tensor:
    """Add N(0, std^2) noise."""
    if std <= 0:
        return features
    return features + torch.randn_like(features) * std


def compose(*fns):
    active = [fn for fn in fns if fn is not None]

    def _apply(x: torch.Tensor) -> torch.Tensor:
        for fn in active:
            x = fn(x)
        return x
    return _apply


def _span(n: int, lo_ratio: float, hi_ratio: float):
    length = max(1, min(int(n * random.uniform(lo_ratio, hi_ratio)), n - 1))
    start = random.randint(0, n - length)
    return start, length


def time_mask(features, max_mask_ratio=0.2, min_mask_ratio=0.05):
    """SpecAugment time masking: zero one contiguous span of frames (same span for the whole batch)."""
    start, length = _span(features.shape[1], min_mask_ratio, max_mask_ratio)
    out = features.clone()
    out[:, start:start + length, :] = 0
    return out


def feature_mask(features, max_mask_ratio=0.1, min_mask_ratio=0.02):
    """SpecAugment feature masking: zero one contiguous span of feature dims."""
    start, length = _span(features.shape[2], min_mask_ratio, max_mask_ratio)
    out = features.clone()
    out[:, :, start:start + length] = 0
    return out


def spec_augment(features, time_mask_ratio=0.2, feature_mask_ratio=0.1, apply_time_mask=True,
                 apply_feature_mask=False):
    if apply_time_mask:
        features = time_mask(features, max_mask_ratio=time_mask_ratio)
    if apply_feature_mask:
        features = feature_mask(features, max_mask_ratio=feature_mask_ratio)
    return features


class FusedAugment:
    """The same augmentation pipeline as `compose(spec_augment, time_shift, channel_drop, gaussian_jitter)` built by
    train.build_augment_fn (reference order, src/train.py:271-289), applied by ONE HIP pass over the batch
    (`dfa_augment_batch`) instead of up to four torch ops.  The per-batch parameters are drawn on the host with the same
    calls, in the same order, as the op-by-op functions above -- seeded runs pick the same spans, shift and per-dim keep
    mask as the reference; only the jitter noise comes from the library's Philox stream (statistically N(0, std^2)).

    fold=True / "cnn2d" / "cnn1d" (classifier training): nothing is computed here at all -- the drawn parameters are ARMED on
    the device context (`dfa_cnn2d_set_train_augment` / `dfa_cnn1d_set_train_augment`) and the batch is returned unchanged;
    the next train-mode forward of that model (and its backward) read x through the same element formula inside the kernels
    that load x (three for the CNN2D, two for the CNN1D), so no augmented copy of the batch is ever written or re-read
    (SURVEY.md section 8(f)3).  The result is identical to the stand-alone pass.

    CUDA tensors only (the product path has no CPU fallback); for CPU tensors use the functions above."""

    def __init__(self, spec_augment=False, time_mask_ratio=0.2, feature_mask=False, feature_mask_ratio=0.1,
                 time_shift=False, time_shift_ratio=0.1, channel_drop=False, channel_drop_prob=0.1,
                 gaussian_jitter=False, gaussian_jitter_std=0.01, out_dtype=None, seed=None, rng_device=None, fold=False):
        self.fold = {True: "cnn2d", False: None, None: None}.get(fold, fold)    # which model's loads take the augmentation
        if self.fold not in (None, "cnn2d", "cnn1d"):
            raise ValueError(f"fold must be False, True, 'cnn2d' or 'cnn1d' (got {fold!r})")
        self.rng_device = rng_device      # where the keep mask is drawn: None = the batch's device (src/augmentation.py:52), 'cpu' = the host generator
        self.spec, self.tm_ratio = bool(spec_augment), float(time_mask_ratio)
        self.fmask, self.fm_ratio = bool(feature_mask), float(feature_mask_ratio)
        self.shift, self.shift_ratio = bool(time_shift), float(time_shift_ratio)
        self.cdrop, self.cdrop_p = bool(channel_drop), float(channel_drop_prob)
        self.jitter, self.jitter_std = bool(gaussian_jitter), float(gaussian_jitter_std)
        self.out_dtype = out_dtype
        self.seed = int(seed) if seed is not None else int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self.calls = 0

    def draw(self, B, T, F, device):
        """(shift, keep[F] or None, tmask_start, tmask_len, fmask_start, fmask_len) in the reference's draw order."""
        tm = fm = (0, 0)
        if self.spec:                                       # spec_augment: time mask, then feature mask
            tm = _span(T, 0.05, self.tm_ratio)
            if self.fmask:
                fm = _span(F, 0.02, self.fm_ratio)
        shift = 0
        if self.shift and self.shift_ratio > 0 and T > 1 and int(T * self.shift_ratio) >= 1:
            limit = int(T * self.shift_ratio)
            shift = random.randint(-limit, limit)
        keep = None
        if self.cdrop and self.cdrop_p > 0:
            keep = (torch.rand((1, 1, F), device=self.rng_device or device) >= self.cdrop_p).to(torch.float32).reshape(F)
            keep = keep.to(device).contiguous()
        return shift, keep, tm[0], tm[1], fm[0], fm[1]

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        import ctypes as C
        from . import _lib
        if x.device.type != "cuda":
            raise RuntimeError("FusedAugment runs on the GPU only; use the op-by-op functions for CPU tensors")
        B, T, F = x.shape
        shift, keep, ts, tl, fs, fl = self.draw(B, T, F, x.device)
        std = self.jitter_std if (self.jitter and self.jitter_std > 0) else 0.0
        if self.fold:
            ctx = _lib.Context.get(x.device)
            self._keep = keep                                   # must outlive the backward pass
            arm = ctx.lib.dfa_cnn2d_set_train_augment if self.fold == "cnn2d" else ctx.lib.dfa_cnn1d_set_train_augment
            _lib.check(ctx.handle, arm(
                ctx.handle, 1, T, F, int(shift), C.c_void_p(keep.data_ptr() if keep is not None else None), ts, tl, fs, fl,
                std, self.seed, self.calls * (B * T * F + 64)))
            self.calls += 1
            return x
        out = torch.empty_strided((B, T, F), x.stride(), dtype=self.out_dtype or x.dtype, device=x.device) \
            if _dense_strides(x) else torch.empty((B, T, F), dtype=self.out_dtype or x.dtype, device=x.device)
        ctx = _lib.Context.get(x.device)
        with torch.cuda.device(ctx.index):
            ctx.use_current_stream()
            code = ctx.lib.dfa_augment_batch(
                ctx.handle, C.c_void_p(x.data_ptr()), _lib.x_dtype_code(x), B, T, F, *x.stride(),
                C.c_void_p(out.data_ptr()), _lib.x_dtype_code(out), *out.stride(), int(shift),
                C.c_void_p(keep.data_ptr() if keep is not None else None), ts, tl, fs, fl, std, self.seed,
                self.calls * (B * T * F + 64))
            _lib.check(ctx.handle, code)
        self.calls += 1
        return out


def _dense_strides(x: torch.Tensor) -> bool:
    """True when x's strides are a permutation of a dense layout (so the output can reuse them, e.g. the [B,T,F] view of
    stored [B,F,T] features)."""
    sizes, strides = list(x.shape), list(x.stride())
    order = sorted(range(len(sizes)), key=lambda i: strides[i])
    expect = 1
    for i in order:
        if strides[i] != expect:
            return False
        expect *= sizes[i]
    return True
