// augment.hip -- the train-time feature augmentations of src/augmentation.py as ONE pass over the batch
// (SURVEY.md section 8 (f) 3): the reference composes up to four full-tensor torch ops per step (src/train.py:68-69,
// src/augmentation.py:5-186): SpecAugment time / feature mask -> circular time shift -> feature-dim drop -> Gaussian jitter.
//   out[b][t][f] = keep[f] * mask(x[b][(t - shift) mod T][f]) + std * N(0,1)
// The random PARAMETERS (mask spans, shift, per-dim keep mask) are drawn on the host exactly as the reference draws them
// (one draw per batch, Python `random` / the torch generator); only the jitter noise comes from the in-kernel Philox
// stream (one normal per element, Box-Muller) -- a torch.randn_like stream cannot be reproduced on another device anyway.
// The element formula lives in rng.h (AugCfg / aug_apply): the training kernels that read x fold the SAME function into
// their loads (dfa_cnn2d_set_train_augment), so this stand-alone pass and the folded form give identical batches.
#include "dfa_internal.h"
#include "rng.h"

namespace dfa {

struct AugArgs {
  const void* x; void* out;
  long long sxb, sxt, sxf, sob, sot, sof;   // element strides of input and output
  int B;
  AugCfg cfg;                               // rng.h: the same definition the training kernels fold into their loads
};

__device__ __forceinline__ float aug_ld(const float* p) { return *p; }
__device__ __forceinline__ float aug_ld(const bf16_t* p) { return bf16_to_float(*p); }
__device__ __forceinline__ void aug_st(float* p, float v) { *p = v; }
__device__ __forceinline__ void aug_st(bf16_t* p, float v) { *p = float_to_bf16(v); }

// thread = 4 consecutive elements of the output's contiguous dimension (TFAST: that is t, else f)
template <typename TI, typename TO, bool TFAST>
__global__ __launch_bounds__(256) void augment_kernel(AugArgs a) {
  const int T = a.cfg.T, F = a.cfg.F;
  const int inner = TFAST ? T : F, outer = TFAST ? F : T;
  const int ngrp = (inner + 3) >> 2;
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)a.B * outer * ngrp;
  if (gid >= total) return;
  const int g = (int)(gid % ngrp);
  const long long rest = gid / ngrp;
  const int o = (int)(rest % outer), b = (int)(rest / outer);
  const TI* xb = (const TI*)a.x + (long long)b * a.sxb;
  TO* ob = (TO*)a.out + (long long)b * a.sob;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int i = 4 * g + e;
    if (i >= inner) break;
    const int t = TFAST ? i : o, f = TFAST ? o : i;
    const int ts = aug_src_t(a.cfg, t);
    const float xraw = aug_ld(xb + (long long)ts * a.sxt + (long long)f * a.sxf);
    aug_st(ob + (long long)t * a.sot + (long long)f * a.sof, aug_apply(a.cfg, xraw, b, t, f));
  }
}

hipError_t launch_augment(const AugArgs& a, int x_dtype, int out_dtype, hipStream_t s) {
  const bool tfast = (a.sot == 1);
  const int inner = tfast ? a.cfg.T : a.cfg.F, outer = tfast ? a.cfg.F : a.cfg.T;
  const long long total = (long long)a.B * outer * ((inner + 3) >> 2);
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define DFA_AUG(TI, TO)                                                                        \
  do {                                                                                         \
    if (tfast) hipLaunchKernelGGL((augment_kernel<TI, TO, true>), grid, block, 0, s, a);       \
    else hipLaunchKernelGGL((augment_kernel<TI, TO, false>), grid, block, 0, s, a);            \
  } while (0)
  if (x_dtype == DFA_DTYPE_BF16) { if (out_dtype == DFA_DTYPE_BF16) DFA_AUG(bf16_t, bf16_t); else DFA_AUG(bf16_t, float); }
  else { if (out_dtype == DFA_DTYPE_BF16) DFA_AUG(float, bf16_t); else DFA_AUG(float, float); }
#undef DFA_AUG
  return hipGetLastError();
}

}  // namespace dfa

using namespace dfa;

extern "C" int dfa_augment_batch(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b,
                                 int64_t stride_t, int64_t stride_f, void* out, int out_dtype, int64_t out_stride_b,
                                 int64_t out_stride_t, int64_t out_stride_f, int shift, const float* keep_f, int tmask_start,
                                 int tmask_len, int fmask_start, int fmask_len, float jitter_std, uint64_t seed,
                                 uint64_t offset) {
  if (!ctx) return DFA_E_NULL_PTR;
  if (!x || !out) return fail(ctx, DFA_E_NULL_PTR, "x and out must be non-null");
  if (x == out) return fail(ctx, DFA_E_UNSUPPORTED, "dfa_augment_batch is out of place (the time shift reads other frames)");
  if ((x_dtype != DFA_DTYPE_F32 && x_dtype != DFA_DTYPE_BF16) || (out_dtype != DFA_DTYPE_F32 && out_dtype != DFA_DTYPE_BF16))
    return fail(ctx, DFA_E_BAD_DTYPE, "dtypes must be fp32 or bf16 (got %d -> %d)", x_dtype, out_dtype);
  if (B < 1 || T < 1 || F < 1) return fail(ctx, DFA_E_BAD_SHAPE, "bad shape [%d,%d,%d]", B, T, F);
  if (tmask_len < 0 || fmask_len < 0 || tmask_start < 0 || fmask_start < 0 || tmask_start + tmask_len > T ||
      fmask_start + fmask_len > F)
    return fail(ctx, DFA_E_BAD_SHAPE, "mask span outside the batch");
  if (!(jitter_std >= 0.f)) return fail(ctx, DFA_E_BAD_SHAPE, "jitter std must be >= 0");
  AugArgs a{};
  a.x = x; a.out = out;
  a.sxb = stride_b; a.sxt = stride_t; a.sxf = stride_f;
  a.sob = out_stride_b; a.sot = out_stride_t; a.sof = out_stride_f;
  a.B = B;
  a.cfg.on = 1; a.cfg.T = T; a.cfg.F = F; a.cfg.shift = ((shift % T) + T) % T; a.cfg.keep = keep_f;
  a.cfg.tm_start = tmask_start; a.cfg.tm_len = tmask_len; a.cfg.fm_start = fmask_start; a.cfg.fm_len = fmask_len;
  a.cfg.std = jitter_std; a.cfg.seed = seed; a.cfg.offset = offset;
  DFA_HIP_CHECK(ctx, launch_augment(a, x_dtype, out_dtype, ctx->stream));
  return DFA_OK;
}
