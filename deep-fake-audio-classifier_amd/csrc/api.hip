// api.hip -- the C ABI of libdfa_hip.so (include/dfa_hip.h): context lifecycle, weight preparation and the
// eval-mode forward of the reference's CNN2D (src/model.py:33-42) as four stream-ordered launches:
//   conv1 (+BN+ReLU+pool)  ->  block 2 MFMA conv (+BN+ReLU+pool)  ->  block 3 MFMA conv (+BN+ReLU+mean_T)  ->  linear.
#include <stdlib.h>

#include <algorithm>

#include "dfa_internal.h"
#include "trace.h"
#include "convt2x2_mfma.h"

using namespace dfa;

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

constexpr int kMaxSeg = 4;

struct Cnn2dPlan {
  int H1, H2;               // rows after pool 1 / pool 2
  size_t a1_off, a2_off, emb_off, total;
};

// forced_split: the context's "time_split" option (> 0 forces a split whatever the batch, so the chunk slabs must exist)
Cnn2dPlan plan_cnn2d(int B, int T, int F, int prec, int forced_split = -1) {
  Cnn2dPlan p;
  const size_t es = (prec == DFA_PREC_BF16) ? 2 : 4;   // fp32, or a hi + lo bf16 pair (DFA_PREC_BF16X3)
  p.H1 = T / 2;
  p.H2 = p.H1 / 2;
  size_t off = 0;
  p.a1_off = off;
  off = align_up(off + (size_t)B * p.H1 * F * 32 * es, 256);
  p.a2_off = off;
  off = align_up(off + (size_t)B * p.H2 * F * 64 * es, 256);
  p.emb_off = off;
  // small batches split the time axis over up to kMaxSeg workgroups per strip: one partial embedding per segment
  const int nstrips = (F + 31) / 32;
  const size_t nemb = (forced_split > 0 || (size_t)B * nstrips < 512) ? kMaxSeg + 1 : 1;
  off = align_up(off + nemb * (size_t)B * 128 * F * sizeof(float), 256);
  p.total = off;
  return p;
}

// iterations per time-axis segment for a kernel that walks `niter` iterations with `nwg` workgroups and `slots` resident
// workgroup slots on the chip: 0 = do not split.  A multiple of `period` (ring / window-buffer phase of the kernel).
int seg_iters_for(int niter, int nwg, int slots, int period, int forced) {
  if (forced == 0 || niter <= period) return 0;
  int seg;
  if (forced > 0) seg = (niter + forced - 1) / forced;
  else {
    if (nwg >= slots) return 0;
    seg = (int)(((long long)niter * nwg + slots - 1) / slots);
  }
  seg = (seg + period - 1) / period * period;
  if (seg >= niter) return 0;
  if ((niter + seg - 1) / seg > kMaxSeg) seg = ((niter + kMaxSeg - 1) / kMaxSeg + period - 1) / period * period;
  return seg >= niter ? 0 : seg;
}

struct Cnn1dPlan {
  size_t h1_off, h2_off, pooled_off, total;
};

Cnn1dPlan plan_cnn1d(int B, int T) {
  Cnn1dPlan p;
  size_t off = 0;
  p.h1_off = off;
  off = align_up(off + (size_t)B * 32 * T * sizeof(float), 256);
  p.h2_off = off;
  off = align_up(off + (size_t)B * 64 * T * sizeof(float), 256);
  p.pooled_off = off;
  off = align_up(off + (size_t)B * 128 * sizeof(float), 256);
  p.total = off;
  return p;
}

struct CaePlan {
  int H[5], W[5];        // H[0]=T, W[0]=F, then after each 2x2 pool
  int Hd[4], Wd[4];      // decoder outputs d1, d2, d3 and recon rows/cols
  size_t e_off[4], raw_off, d_off[3], part_off, total;
  int nblk;
  bool ok;
};

CaePlan plan_cae(int B, int T, int F, int prec) {
  CaePlan p;
  const size_t es = (prec == DFA_PREC_BF16) ? 2 : 4;
  p.H[0] = T; p.W[0] = F;
  for (int l = 1; l <= 4; ++l) { p.H[l] = p.H[l - 1] / 2; p.W[l] = p.W[l - 1] / 2; }
  p.Hd[0] = 2 * p.H[4]; p.Wd[0] = 2 * p.W[4];
  p.Hd[1] = 2 * p.Hd[0]; p.Wd[1] = 2 * p.Wd[0] + 1;   // output_padding = (0, 1)
  p.Hd[2] = 2 * p.Hd[1]; p.Wd[2] = 2 * p.Wd[1];
  p.Hd[3] = 2 * p.Hd[2]; p.Wd[3] = 2 * p.Wd[2];
  p.ok = (p.H[4] >= 1 && p.W[4] >= 1 && p.Wd[3] == F);
  const int ch[4] = {32, 64, 128, 256};
  size_t off = 0;
  for (int l = 0; l < 4; ++l) { p.e_off[l] = off; off = align_up(off + (size_t)B * p.H[l + 1] * p.W[l + 1] * ch[l] * es, 256); }
  p.raw_off = off;
  if (prec == DFA_PREC_F32) off = align_up(off + (size_t)B * p.H[3] * p.W[3] * 256 * 4, 256);
  const int dch[3] = {128, 64, 32};
  for (int l = 0; l < 3; ++l) { p.d_off[l] = off; off = align_up(off + (size_t)B * p.Hd[l] * p.Wd[l] * dch[l] * es, 256); }
  p.nblk = cae_dec4_blocks(T, p.Wd[2]);
  p.part_off = off;
  off = align_up(off + (size_t)B * p.nblk * sizeof(float), 256);
  p.total = off;
  return p;
}

}  // namespace

extern "C" {

int dfa_version(void) { return DFA_VERSION; }

const char* dfa_error_name(int code) {
  switch (code) {
    case DFA_OK: return "DFA_OK";
    case DFA_E_BAD_SHAPE: return "DFA_E_BAD_SHAPE";
    case DFA_E_BAD_DTYPE: return "DFA_E_BAD_DTYPE";
    case DFA_E_NULL_PTR: return "DFA_E_NULL_PTR";
    case DFA_E_NOT_PREPARED: return "DFA_E_NOT_PREPARED";
    case DFA_E_HIP: return "DFA_E_HIP";
    case DFA_E_WORKSPACE: return "DFA_E_WORKSPACE";
    case DFA_E_UNSUPPORTED: return "DFA_E_UNSUPPORTED";
    default: return "DFA_E_UNKNOWN";
  }
}

int dfa_ctx_create(int device_id, void* hip_stream, dfa_ctx** out) {
  if (!out) return DFA_E_NULL_PTR;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return DFA_E_HIP;
  if (hipSetDevice(device_id) != hipSuccess) return DFA_E_HIP;
  dfa_ctx* c = new dfa_ctx();
  c->device = device_id;
  c->stream = (hipStream_t)hip_stream;
  if (hipMalloc(&c->zero_page, 256) != hipSuccess || hipMemset(c->zero_page, 0, 256) != hipSuccess) {
    delete c;
    return DFA_E_HIP;
  }
  const char* dma = getenv("DFA_CONV_DMA");   // unset = per-kernel choice measured on MI355X (see launch_cnn2d_block*)
  c->conv_dma = (dma && (dma[0] == '0' || dma[0] == '1')) ? dma[0] - '0' : -1;
  *out = c;
  return DFA_OK;
}

int dfa_ctx_destroy(dfa_ctx* ctx) {
  if (!ctx) return DFA_E_NULL_PTR;
  (void)hipSetDevice(ctx->device);
  if (ctx->cnn2d.packed) (void)hipFree(ctx->cnn2d.packed);
  if (ctx->cnn2d.train_packed) (void)hipFree(ctx->cnn2d.train_packed);
  if (ctx->cnn1d.packed) (void)hipFree(ctx->cnn1d.packed);
  if (ctx->cnn1d.train_packed) (void)hipFree(ctx->cnn1d.train_packed);
  if (ctx->cae.packed) (void)hipFree(ctx->cae.packed);
  if (ctx->cae.train_packed) (void)hipFree(ctx->cae.train_packed);
  if (ctx->zero_page) (void)hipFree(ctx->zero_page);
  if (ctx->clock_buf) (void)hipFree(ctx->clock_buf);
  if (ctx->mse_partial) (void)hipFree(ctx->mse_partial);
  if (ctx->aug_keep) (void)hipFree(ctx->aug_keep);
  for (auto& t : ctx->slots) {
    for (auto e : t.start) (void)hipEventDestroy(e);
    for (auto e : t.stop) (void)hipEventDestroy(e);
  }
  delete ctx;
  return DFA_OK;
}

int dfa_ctx_set_stream(dfa_ctx* ctx, void* hip_stream) {
  if (!ctx) return DFA_E_NULL_PTR;
  ctx->stream = (hipStream_t)hip_stream;
  return DFA_OK;
}

const char* dfa_last_error(const dfa_ctx* ctx) { return ctx ? ctx->err : "null context"; }

namespace {
// Test hook ("poison_lds"): fills the whole 160 KB of LDS of every CU with a 16-bit pattern (0x7fc0 = bf16 NaN, 0xffff = NaN in
// either width).  LDS is not cleared between workgroups, so a kernel that lets bytes it never wrote reach a result shows up in the
// parity tests that run after this instead of once in a few thousand launches.
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned pattern, unsigned* sink) {
  extern __shared__ unsigned lds_words[];
  constexpr int N = 160 * 1024 / 4;
  for (int i = threadIdx.x; i < N; i += 256) lds_words[i] = pattern;
  __syncthreads();
  __builtin_amdgcn_s_sleep(127);
  if (lds_words[(threadIdx.x * 97 + blockIdx.x) % N] != pattern) sink[0] = 1;   // keeps the stores; never true
}
hipError_t launch_poison_lds(dfa_ctx* ctx, unsigned half) {
  hipError_t e = hipSetDevice(ctx->device);   // the attribute is per device: set it on every call (a test hook, cheap)
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)poison_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  half &= 0xffffu;
  hipLaunchKernelGGL(poison_lds_kernel, dim3(2048), dim3(256), 160 * 1024, ctx->stream, half | (half << 16), (unsigned*)ctx->zero_page);
  return hipGetLastError();
}
}  // namespace

int dfa_ctx_set_option(dfa_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return DFA_E_NULL_PTR;
  if (strcmp(name, "poison_lds") == 0) {
    hipError_t e = launch_poison_lds(ctx, (unsigned)value);
    return e == hipSuccess ? DFA_OK : fail(ctx, DFA_E_HIP, "poison_lds: %s", hipGetErrorString(e));
  }
  if (strcmp(name, "train_conv_variant") == 0) { set_train_conv_variant(value); return DFA_OK; }
  if (strcmp(name, "wgrad_variant") == 0) { set_wgrad_variant(value); return DFA_OK; }
  if (strcmp(name, "time_split") == 0) { ctx->time_split = value; return DFA_OK; }
  if (strcmp(name, "clock_probe") == 0) {
    if (value && !ctx->clock_buf) {
      DFA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
      DFA_HIP_CHECK(ctx, hipMalloc((void**)&ctx->clock_buf, 1024 * 2 * sizeof(long long)));
    }
    if (value) DFA_HIP_CHECK(ctx, hipMemsetAsync(ctx->clock_buf, 0, 1024 * 2 * sizeof(long long), ctx->stream));
    ctx->clock_probe = value ? 1 : 0;
    return DFA_OK;
  }
  if (strcmp(name, "conv1_bwd_fused") == 0) { ctx->conv1_bwd_fused = value ? 1 : 0; return DFA_OK; }
  if (strcmp(name, "dgrad_m16") == 0) { ctx->dgrad_m16 = value ? 1 : 0; return DFA_OK; }
  if (strcmp(name, "conv1_mfma") == 0) { ctx->conv1_mfma = value ? 1 : 0; return DFA_OK; }
  if (strcmp(name, "cae_enc1_mfma") == 0) { ctx->cae_enc1_mfma = value ? 1 : 0; return DFA_OK; }
  if (strcmp(name, "cae_enc_dma") == 0) { ctx->cae_enc_dma = value ? 1 : 0; return DFA_OK; }
  if (strcmp(name, "cae_dgrad_mfma") == 0) { ctx->cae_dgrad_mfma = value ? 1 : 0; return DFA_OK; }
  if (strcmp(name, "cae_conv_stats") == 0) { ctx->cae_conv_stats = value ? 1 : 0; return DFA_OK; }
  if (strcmp(name, "cae_bwd_fold") == 0) { ctx->cae_bwd_fold = value ? 1 : 0; return DFA_OK; }
  if (strcmp(name, "cae_enc4_wide") == 0) { ctx->cae_enc4_wide = value ? 1 : 0; return DFA_OK; }
  if (strcmp(name, "cae_dec_fused") == 0) { ctx->cae_dec_fused = value ? 1 : 0; return DFA_OK; }
  if (strcmp(name, "cnn1d_train_x3") == 0) { ctx->cnn1d_train_x3 = (value < 0 || value > 3) ? 1 : value; return DFA_OK; }
  if (strcmp(name, "cnn1d_fused") == 0) { ctx->cnn1d_fused = value < 0 ? 0 : (value > 2 ? 1 : value); return DFA_OK; }
  if (strcmp(name, "block3_m16") == 0) { ctx->block3_m16 = value ? 1 : 0; return DFA_OK; }
  if (strcmp(name, "fuse_conv1") == 0) { ctx->fuse_conv1 = value ? 1 : 0; return DFA_OK; }
  if (strcmp(name, "lds_pipe") == 0) { ctx->lds_pipe = value ? 1 : 0; return DFA_OK; }
  if (strcmp(name, "conv_dma") == 0) { ctx->conv_dma = value < 0 ? -1 : value; return DFA_OK; }
  return fail(ctx, DFA_E_UNSUPPORTED, "unknown option '%s'", name);
}

int dfa_ctx_timing_enable(dfa_ctx* ctx, int enable) {
  if (!ctx) return DFA_E_NULL_PTR;
  ctx->timing = (enable == 1) ? 0xffffffffu : (unsigned)enable;   // 1 = all slots, otherwise a bit mask of slots
  return DFA_OK;
}

int dfa_ctx_timing_reset(dfa_ctx* ctx) {
  if (!ctx) return DFA_E_NULL_PTR;
  for (auto& t : ctx->slots) t.used = 0;
  return DFA_OK;
}

int dfa_ctx_timing_read(dfa_ctx* ctx, int slot, float* total_ms, int* count) {
  if (!ctx || !total_ms || !count) return DFA_E_NULL_PTR;
  if (slot < 0 || slot >= kMaxSlots) return fail(ctx, DFA_E_BAD_SHAPE, "timing slot %d out of range", slot);
  SlotTimer& t = ctx->slots[slot];
  float sum = 0.f;
  for (int i = 0; i < t.used; ++i) {
    DFA_HIP_CHECK(ctx, hipEventSynchronize(t.stop[i]));
    float ms = 0.f;
    DFA_HIP_CHECK(ctx, hipEventElapsedTime(&ms, t.start[i], t.stop[i]));
    sum += ms;
  }
  *total_ms = sum;
  *count = t.used;
  return DFA_OK;
}

int dfa_ctx_clock_read(dfa_ctx* ctx, double* ghz_median, double* ghz_min, double* ghz_max, int* workgroups) {
  if (!ctx || !ghz_median || !workgroups) return DFA_E_NULL_PTR;
  *ghz_median = 0.0; *workgroups = 0;
  if (ghz_min) *ghz_min = 0.0;
  if (ghz_max) *ghz_max = 0.0;
  if (!ctx->clock_buf) return fail(ctx, DFA_E_NOT_PREPARED, "set the context option clock_probe = 1 and run a bf16 CNN2D forward first");
  DFA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  DFA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  static thread_local long long h[1024 * 2];
  DFA_HIP_CHECK(ctx, hipMemcpy(h, ctx->clock_buf, sizeof(h), hipMemcpyDeviceToHost));
  double g[1024];
  int n = 0;
  for (int i = 0; i < 1024; ++i)
    if (h[2 * i + 1] > 0 && h[2 * i] > 0) g[n++] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0);   // cycles / (ticks * 10 ns) = GHz
  if (n == 0) return DFA_OK;
  std::sort(g, g + n);
  *ghz_median = g[n / 2];
  if (ghz_min) *ghz_min = g[0];
  if (ghz_max) *ghz_max = g[n - 1];
  *workgroups = n;
  return DFA_OK;
}

int dfa_ctx_debug_read(dfa_ctx* ctx, long long* host_words, int n) {
  if (!ctx || !host_words) return DFA_E_NULL_PTR;
  if (n < 0 || n > 2048) return fail(ctx, DFA_E_BAD_SHAPE, "the probe buffer holds 2048 words (got %d)", n);
  if (!ctx->clock_buf) return fail(ctx, DFA_E_NOT_PREPARED, "set the context option clock_probe = 1 first");
  DFA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  DFA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  DFA_HIP_CHECK(ctx, hipMemcpy(host_words, ctx->clock_buf, (size_t)n * sizeof(long long), hipMemcpyDeviceToHost));
  return DFA_OK;
}

const char* dfa_dominant_kernel(int model, int precision) {
  if (model == DFA_MODEL_CNN2D)
    return precision == DFA_PREC_BF16 ? "conv3_m16_meant_kernel" : precision == DFA_PREC_BF16X3 ? "conv_split_kernel" : "conv3x3_mfma_kernel";
  return "";
}

/* ------------------------------------------------------------------------------------------------ CNN2D */
int dfa_cnn2d_set_params(dfa_ctx* ctx, const float* const* device_params, int n, int in_features, int base_channels) {
  if (!ctx || !device_params) return DFA_E_NULL_PTR;
  if (n != DFA_CNN2D_NPARAMS) return fail(ctx, DFA_E_BAD_SHAPE, "cnn2d expects %d parameter pointers, got %d", DFA_CNN2D_NPARAMS, n);
  if (base_channels != 32) return fail(ctx, DFA_E_UNSUPPORTED, "cnn2d HIP path is built for base_channels=32 (got %d)", base_channels);
  if (in_features < 1) return fail(ctx, DFA_E_BAD_SHAPE, "in_features must be positive (got %d)", in_features);
  for (int i = 0; i < n; ++i)
    if (!device_params[i]) return fail(ctx, DFA_E_NULL_PTR, "cnn2d parameter %d is null", i);
  for (int i = 0; i < n; ++i) ctx->cnn2d.p[i] = device_params[i];
  ctx->cnn2d.in_features = in_features;
  ctx->cnn2d.have_params = true;
  ctx->cnn2d.prepared_prec = -1;
  return DFA_OK;
}

int dfa_cnn2d_prepare(dfa_ctx* ctx, int precision) {
  if (!ctx) return DFA_E_NULL_PTR;
  Cnn2dState& m = ctx->cnn2d;
  if (!m.have_params) return fail(ctx, DFA_E_NOT_PREPARED, "dfa_cnn2d_set_params has not been called");
  if (precision != DFA_PREC_F32 && precision != DFA_PREC_BF16 && precision != DFA_PREC_BF16X3)
    return fail(ctx, DFA_E_BAD_DTYPE, "unknown precision %d", precision);
  DFA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // one allocation: w1[288] b1[32] | bias2[64] bias3[128] | wpack2 | wpack3   (sized for fp32, the larger mode)
  const size_t w2_bytes = (size_t)64 * 32 * 9 * 4, w3_bytes = (size_t)128 * 64 * 9 * 4;
  const size_t m16_bytes = (size_t)128 * 64 * 9 * 2;   // block-3 image for the 16x16x32 kernel (bf16)
  const size_t c1_bytes = 4 * 64 * 16 + 256 + m16_bytes;   // block-1 MFMA operands + bias of the fused kernel
  const size_t need = align_up((288 + 32 + 64 + 128) * sizeof(float), 256) + w2_bytes + w3_bytes + c1_bytes;
  if (!m.packed) {
    DFA_HIP_CHECK(ctx, hipMalloc(&m.packed, need));
    m.packed_bytes = need;
  }
  char* base = (char*)m.packed;
  m.w1 = (float*)base;
  m.b1 = m.w1 + 288;
  m.c2.bias = m.b1 + 32;
  m.c3.bias = m.c2.bias + 64;
  char* wp = base + align_up((288 + 32 + 64 + 128) * sizeof(float), 256);
  m.c2.wpack = (uint4*)wp;
  m.c3.wpack = (uint4*)(wp + w2_bytes);
  m.c1pack = (uint4*)(wp + w2_bytes + w3_bytes);
  m.c1bias = (float*)(wp + w2_bytes + w3_bytes + 4 * 64 * 16);
  m.c3_m16 = (uint4*)(wp + w2_bytes + w3_bytes + 4 * 64 * 16 + 256);
  const float* const* p = m.p;
  DFA_HIP_CHECK(ctx, launch_fold_conv1(p[0], p[1], p[2], p[3], p[4], p[5], m.w1, m.b1, 32, ctx->stream));
  DFA_HIP_CHECK(ctx, launch_pack_conv1_mfma(m.w1, m.b1, m.c1pack, m.c1bias, ctx->stream));
  if (precision == DFA_PREC_BF16X3) {
    // hi/lo split images (conv_split.hip): the same byte counts as the fp32 images, so they live in the same regions
    DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3_split(p[6], p[7], p[8], p[9], p[10], p[11], 32, 64, m.c2.wpack, m.c2.bias, 0.5f, ctx->stream));
    DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3_split(p[12], p[13], p[14], p[15], p[16], p[17], 64, 128, m.c3.wpack, m.c3.bias, 1.0f, ctx->stream));
    m.prepared_prec = precision;
    return DFA_OK;
  }
  DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3(p[6], p[7], p[8], p[9], p[10], p[11], 32, 0, 32, 64, precision, m.c2.wpack, m.c2.bias, ctx->stream, 1, 0.5f));
  DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3(p[12], p[13], p[14], p[15], p[16], p[17], 64, 0, 64, 128, precision, m.c3.wpack, m.c3.bias, ctx->stream));
  if (precision == DFA_PREC_BF16)
    DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3_m16(p[12], p[13], p[14], p[15], p[16], p[17], 64, 128, m.c3_m16, ctx->stream));
  m.prepared_prec = precision;
  return DFA_OK;
}

size_t dfa_workspace_bytes(const dfa_ctx* ctx, int model, int B, int T, int F, int precision) {
  if (B < 1 || T < 1 || F < 1) return 0;
  if (model == DFA_MODEL_CNN2D) return plan_cnn2d(B, T, F, precision, ctx ? ctx->time_split : -1).total;
  if (model == DFA_MODEL_CNN1D) return plan_cnn1d(B, T).total;
  if (model == DFA_MODEL_CAE) return plan_cae(B, T, F, precision).total;
  return 0;
}

int dfa_cnn2d_forward(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b, int64_t stride_t,
                      int64_t stride_f, float* logits, float* embedding, void* workspace, size_t workspace_bytes) {
  TraceRange trace_("dfa_cnn2d_forward");
  if (!ctx) return DFA_E_NULL_PTR;
  Cnn2dState& m = ctx->cnn2d;
  if (m.prepared_prec < 0) return fail(ctx, DFA_E_NOT_PREPARED, "dfa_cnn2d_prepare has not been called since the last set_params");
  if (!x || !logits || !workspace) return fail(ctx, DFA_E_NULL_PTR, "x, logits and workspace must be non-null");
  if (x_dtype != DFA_DTYPE_F32 && x_dtype != DFA_DTYPE_BF16) return fail(ctx, DFA_E_BAD_DTYPE, "x dtype %d not supported", x_dtype);
  if (B < 1) return fail(ctx, DFA_E_BAD_SHAPE, "batch must be >= 1 (got %d)", B);
  if (F != m.in_features)
    return fail(ctx, DFA_E_BAD_SHAPE, "feature dim %d does not match in_features=%d of the classifier (src/model.py:31)", F, m.in_features);
  if (T < 4) return fail(ctx, DFA_E_BAD_SHAPE, "T=%d is too short: two (2,1) average pools need T >= 4", T);
  const int prec = m.prepared_prec;
  const Cnn2dPlan pl = plan_cnn2d(B, T, F, prec, ctx->time_split);
  if (workspace_bytes < pl.total)
    return fail(ctx, DFA_E_WORKSPACE, "workspace too small: %zu < %zu bytes", workspace_bytes, pl.total);
  if (((uintptr_t)workspace & 255) != 0) return fail(ctx, DFA_E_WORKSPACE, "workspace must be 256-byte aligned");
  if (embedding && ((uintptr_t)embedding & 15) != 0)   // emb_reduce_kernel / the block-3 epilogues store 16 bytes at a time
    return fail(ctx, DFA_E_BAD_SHAPE, "embedding must be 16-byte aligned (got %p)", (void*)embedding);
  char* ws = (char*)workspace;
  void* a1 = ws + pl.a1_off;
  void* a2 = ws + pl.a2_off;
  float* emb = embedding ? embedding : (float*)(ws + pl.emb_off);
  int nseg3 = 1;
  hipStream_t s = ctx->stream;
  // bf16 mode: blocks 1 and 2 run as one kernel and a1 stays on chip (timing slot 1); fp32 features are rounded to bf16
  // as the kernel loads them (bf16 storage mode), exactly like a caller-side .to(bfloat16)
  const bool fused12 = prec == DFA_PREC_BF16 && ctx->fuse_conv1;
  // Small batches (B * strips below the chip's resident-workgroup count; the reference's predict.py default is batch 32):
  // every workgroup would walk the whole time axis alone, so the axis is split into segments that separate workgroups walk
  // (blocks 1+2: disjoint output rows; block 3: partial means per segment, added up by the classifier kernel).
  const int nstrips30 = (F + 29) / 30;     // conv12_fused, conv3_m16 and conv_split own 30 columns per strip (conv3x3_mfma 32)
  if (fused12) {
    ScopedSlot ts(ctx, 1);
    const int seg12 = seg_iters_for((pl.H1 + 3) / 4, B * nstrips30, 512, 6, ctx->time_split);
    DFA_HIP_CHECK(ctx, launch_conv12_fused(x, x_dtype, stride_b, stride_t, stride_f, m.c1pack, m.c1bias, m.c2.wpack, m.c2.bias, a2,
                                           B, T, F, s, ctx->lds_pipe, seg12));
  } else {
    ScopedSlot ts(ctx, 0);
    DFA_HIP_CHECK(ctx, launch_conv1(x, x_dtype, stride_b, stride_t, stride_f, m.w1, m.b1, a1, prec, B, T, F, s));
  }
  if (!fused12) {
    ScopedSlot ts(ctx, 1);
    ConvArgs a{};
    a.in = a1; a.wpack = m.c2.wpack; a.bias = m.c2.bias; a.out = a2; a.emb = nullptr;
    a.B = B; a.H = pl.H1; a.W = F; a.COUT = 64; a.inv_h = 0.f; a.relu = 1; a.zero_page = ctx->zero_page;
    if (prec == DFA_PREC_BF16X3) a.seg_iters = seg_iters_for((pl.H1 + 1) / 2, B * nstrips30, 768, 3, ctx->time_split);
    if (prec == DFA_PREC_BF16X3) DFA_HIP_CHECK(ctx, launch_cnn2d_block2_split(a, s, ctx->lds_pipe));
    else DFA_HIP_CHECK(ctx, launch_cnn2d_block2(prec, a, s, ctx->conv_dma, ctx->lds_pipe));
  }
  {
    ScopedSlot ts(ctx, 2);
    ConvArgs a{};
    a.in = a2; a.wpack = m.c3.wpack; a.bias = m.c3.bias; a.out = nullptr; a.emb = emb;
    a.B = B; a.H = pl.H2; a.W = F; a.COUT = 128; a.inv_h = 1.0f / (float)pl.H2; a.relu = 1; a.zero_page = ctx->zero_page;
    const int niter3 = (pl.H2 + 1) / 2;
    const int chunk3 = 6 * std::max(2, (niter3 + 6 * kMaxSeg - 1) / (6 * kMaxSeg));   // canonical chunks: depend on T only (12 iterations = 24 rows for T = 321), <= kMaxSeg of them
    if (prec == DFA_PREC_BF16X3) {
      a.chunk_iters = chunk3;
      a.seg_iters = seg_iters_for(niter3, B * nstrips30, 256, chunk3, ctx->time_split);
    } else if (prec == DFA_PREC_BF16 && ctx->block3_m16) {
      a.chunk_iters = chunk3;
      a.seg_iters = seg_iters_for(niter3, B * nstrips30, 512, chunk3, ctx->time_split);
    }
    if (a.seg_iters) {           // one slab of (unscaled) sums per canonical chunk in the workspace; the classifier kernel adds them
      nseg3 = (niter3 + chunk3 - 1) / chunk3;
      a.emb_seg_stride = (size_t)B * 128 * F;
      a.emb = (float*)(ws + pl.emb_off) + a.emb_seg_stride;     // slab 0 of the region is the reduced embedding
    }
    if (prec == DFA_PREC_BF16X3) {
      DFA_HIP_CHECK(ctx, launch_cnn2d_block3_split(a, s, ctx->lds_pipe));
    } else if (prec == DFA_PREC_BF16 && ctx->block3_m16) {
      a.wpack = m.c3_m16;
      a.clock_stamps = ctx->clock_probe ? ctx->clock_buf : nullptr;
      DFA_HIP_CHECK(ctx, launch_cnn2d_block3_m16(a, s, ctx->lds_pipe));
    } else {
      DFA_HIP_CHECK(ctx, launch_cnn2d_block3(prec, a, s, ctx->conv_dma, ctx->lds_pipe));
    }
  }
  {
    ScopedSlot ts(ctx, 3);
    if (nseg3 > 1) {   // chunk slabs 1 .. nseg3 of the workspace region -> the embedding (slab 0 of the region, or the caller's buffer)
      const size_t n = (size_t)B * 128 * F;
      DFA_HIP_CHECK(ctx, launch_emb_reduce((const float*)(ws + pl.emb_off) + n, nseg3, n, n, 1.0f / (float)pl.H2, emb, s));
    }
    DFA_HIP_CHECK(ctx, launch_linear(emb, m.p[18], m.p[19], logits, B, 128 * F, s));
  }
  return DFA_OK;
}

/* ------------------------------------------------------------------------------------------------ CNN1D */
int dfa_cnn1d_set_params(dfa_ctx* ctx, const float* const* device_params, int n, int in_features, int base_channels) {
  if (!ctx || !device_params) return DFA_E_NULL_PTR;
  if (n != DFA_CNN1D_NPARAMS) return fail(ctx, DFA_E_BAD_SHAPE, "cnn1d expects %d parameter pointers, got %d", DFA_CNN1D_NPARAMS, n);
  if (base_channels != 32) return fail(ctx, DFA_E_UNSUPPORTED, "cnn1d HIP path is built for base_channels=32 (got %d)", base_channels);
  if (in_features < 1) return fail(ctx, DFA_E_BAD_SHAPE, "in_features must be positive (got %d)", in_features);
  for (int i = 0; i < n; ++i)
    if (!device_params[i]) return fail(ctx, DFA_E_NULL_PTR, "cnn1d parameter %d is null", i);
  for (int i = 0; i < n; ++i) ctx->cnn1d.p[i] = device_params[i];
  ctx->cnn1d.in_features = in_features;
  ctx->cnn1d.have_params = true;
  ctx->cnn1d.prepared = false;
  return DFA_OK;
}

int dfa_cnn1d_prepare(dfa_ctx* ctx) {
  if (!ctx) return DFA_E_NULL_PTR;
  Cnn1dState& m = ctx->cnn1d;
  if (!m.have_params) return fail(ctx, DFA_E_NOT_PREPARED, "dfa_cnn1d_set_params has not been called");
  DFA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int cin[3] = {m.in_features, 32, 64}, cout[3] = {32, 64, 128};
  size_t off[3], boff[3], total = 0;
  for (int l = 0; l < 3; ++l) { off[l] = total; total = align_up(total + (size_t)cout[l] * cin[l] * 3 * 4, 256); }
  for (int l = 0; l < 3; ++l) { boff[l] = total; total = align_up(total + (size_t)cout[l] * 4, 256); }
  size_t poff[3], xoff[3];
  for (int l = 0; l < 3; ++l) { poff[l] = total; total = align_up(total + cnn1d_fused_pack_floats(cin[l], cout[l], l) * 4, 256); }
  for (int l = 0; l < 3; ++l) { xoff[l] = total; total = align_up(total + cnn1d_x3_pack_bytes(cin[l], cout[l]), 256); }
  if (m.packed) { DFA_HIP_CHECK(ctx, hipFree(m.packed)); m.packed = nullptr; }
  DFA_HIP_CHECK(ctx, hipMalloc(&m.packed, total));
  for (int l = 0; l < 3; ++l) {
    m.w[l] = (float*)((char*)m.packed + off[l]);
    m.b[l] = (float*)((char*)m.packed + boff[l]);
    const float* const* q = m.p + 6 * l;
    DFA_HIP_CHECK(ctx, launch_fold_conv1d(q[0], q[1], q[2], q[3], q[4], q[5], m.w[l], m.b[l], cin[l], cout[l], ctx->stream));
    m.wp[l] = (float*)((char*)m.packed + poff[l]);
    DFA_HIP_CHECK(ctx, launch_pack_cnn1d_fused(m.w[l], m.wp[l], cin[l], cout[l], l, ctx->stream));
    m.wx[l] = (char*)m.packed + xoff[l];
    DFA_HIP_CHECK(ctx, launch_pack_cnn1d_x3(m.w[l], m.wx[l], cin[l], cout[l], ctx->stream));
  }
  m.prepared = true;
  return DFA_OK;
}

int dfa_cnn1d_forward(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b, int64_t stride_t,
                      int64_t stride_f, float* logits, void* workspace, size_t workspace_bytes) {
  TraceRange trace_("dfa_cnn1d_forward");
  if (!ctx) return DFA_E_NULL_PTR;
  Cnn1dState& m = ctx->cnn1d;
  if (!m.prepared) return fail(ctx, DFA_E_NOT_PREPARED, "dfa_cnn1d_prepare has not been called since the last set_params");
  if (!x || !logits || !workspace) return fail(ctx, DFA_E_NULL_PTR, "x, logits and workspace must be non-null");
  if (x_dtype != DFA_DTYPE_F32) return fail(ctx, DFA_E_BAD_DTYPE, "cnn1d takes float32 input (got dtype %d)", x_dtype);
  if (B < 1 || T < 1) return fail(ctx, DFA_E_BAD_SHAPE, "B and T must be >= 1 (got %d, %d)", B, T);
  if (F != m.in_features)
    return fail(ctx, DFA_E_BAD_SHAPE, "feature dim %d does not match in_features=%d of the first Conv1d (src/model_cnn1d.py:17)", F, m.in_features);
  const Cnn1dPlan pl = plan_cnn1d(B, T);
  if (workspace_bytes < pl.total) return fail(ctx, DFA_E_WORKSPACE, "workspace too small: %zu < %zu bytes", workspace_bytes, pl.total);
  char* ws = (char*)workspace;
  float *h1 = (float*)(ws + pl.h1_off), *h2 = (float*)(ws + pl.h2_off), *pooled = (float*)(ws + pl.pooled_off);
  hipStream_t s = ctx->stream;
  // (the fused kernel addresses an utterance's elements with 32-bit offsets)
  const bool off32 = stride_t >= 0 && stride_f >= 0 && (int64_t)(F - 1) * stride_f + (int64_t)(T - 1) * stride_t < ((int64_t)1 << 31);
  if (ctx->cnn1d_fused == 1 && cnn1d_fused_x3_supports(x, stride_b, stride_t, stride_f, T, F)) {   // split-bf16 matrix cores (slot 4)
    ScopedSlot ts(ctx, 4);
    DFA_HIP_CHECK(ctx, launch_cnn1d_fused_x3((const float*)x, m.wx[0], m.b[0], m.wx[1], m.b[1], m.wx[2], m.b[2], m.p[18], m.p[19], logits, B, T, F,
                                             s, ctx->clock_probe ? ctx->clock_buf : nullptr));
    return DFA_OK;
  }
  if (ctx->cnn1d_fused && cnn1d_fused_supports(T) && off32) {   // one kernel, one global write per utterance (timing slot 4)
    ScopedSlot ts(ctx, 4);
    DFA_HIP_CHECK(ctx, launch_cnn1d_fused((const float*)x, stride_b, stride_t, stride_f, m.wp[0], m.b[0], m.wp[1], m.b[1], m.wp[2], m.b[2],
                                          m.p[18], m.p[19], logits, B, T, F, s, ctx->clock_probe ? ctx->clock_buf : nullptr));
    return DFA_OK;
  }
  { ScopedSlot ts(ctx, 4);
    DFA_HIP_CHECK(ctx, launch_conv1d((const float*)x, stride_b, stride_f, stride_t, m.w[0], m.b[0], h1, B, F, 32, T, false, s)); }
  { ScopedSlot ts(ctx, 5);
    DFA_HIP_CHECK(ctx, launch_conv1d(h1, (int64_t)32 * T, T, 1, m.w[1], m.b[1], h2, B, 32, 64, T, false, s)); }
  { ScopedSlot ts(ctx, 6);
    DFA_HIP_CHECK(ctx, launch_conv1d(h2, (int64_t)64 * T, T, 1, m.w[2], m.b[2], pooled, B, 64, 128, T, true, s)); }
  { ScopedSlot ts(ctx, 7);
    DFA_HIP_CHECK(ctx, launch_linear(pooled, m.p[18], m.p[19], logits, B, 128, s)); }
  return DFA_OK;
}

/* ------------------------------------------------------------------------------------------------ CAE */
int dfa_cae_set_params(dfa_ctx* ctx, const float* const* device_params, int n, int base_channels) {
  if (!ctx || !device_params) return DFA_E_NULL_PTR;
  if (n != DFA_CAE_NPARAMS) return fail(ctx, DFA_E_BAD_SHAPE, "cae expects %d parameter pointers, got %d", DFA_CAE_NPARAMS, n);
  if (base_channels != 32) return fail(ctx, DFA_E_UNSUPPORTED, "cae HIP path is built for base_channels=32 (got %d)", base_channels);
  for (int i = 0; i < n; ++i)
    if (!device_params[i]) return fail(ctx, DFA_E_NULL_PTR, "cae parameter %d is null", i);
  for (int i = 0; i < n; ++i) ctx->cae.p[i] = device_params[i];
  ctx->cae.have_params = true;
  ctx->cae.prepared_prec = -1;
  return DFA_OK;
}

int dfa_cae_prepare(dfa_ctx* ctx, int precision) {
  if (!ctx) return DFA_E_NULL_PTR;
  CaeState& m = ctx->cae;
  if (!m.have_params) return fail(ctx, DFA_E_NOT_PREPARED, "dfa_cae_set_params has not been called");
  if (precision != DFA_PREC_F32 && precision != DFA_PREC_BF16) return fail(ctx, DFA_E_BAD_DTYPE, "unknown precision %d", precision);
  DFA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int ecin[3] = {32, 64, 128}, ecout[3] = {64, 128, 256};
  const int dcin[3] = {256, 128, 64}, dcout[3] = {128, 64, 32};
  size_t off = align_up((288 + 32) * sizeof(float), 256);
  size_t eb[3], ew[3], db[3], dw[3];
  for (int l = 0; l < 3; ++l) { eb[l] = off; off = align_up(off + ecout[l] * 4, 256); }
  for (int l = 0; l < 3; ++l) { db[l] = off; off = align_up(off + dcout[l] * 4, 256); }
  for (int l = 0; l < 3; ++l) { ew[l] = off; off = align_up(off + (size_t)ecout[l] * ecin[l] * 9 * 4, 256); }
  for (int l = 0; l < 3; ++l) { dw[l] = off; off = align_up(off + (size_t)dcout[l] * dcin[l] * 4 * 4, 256); }
  const size_t cst_off = off;
  off = align_up(off + 16 * sizeof(float), 256);
  const size_t c1p_off = off;
  off = align_up(off + 6 * 64 * 16 + 256, 256);
  const size_t d4p_off = off;
  off = align_up(off + 4 * 64 * 16, 256);
  if (!m.packed) DFA_HIP_CHECK(ctx, hipMalloc(&m.packed, off));
  char* base = (char*)m.packed;
  m.w1 = (float*)base;
  m.b1 = m.w1 + 288;
  for (int l = 0; l < 3; ++l) {
    m.enc[l].bias = (float*)(base + eb[l]); m.enc[l].wpack = (uint4*)(base + ew[l]);
    m.dec[l].bias = (float*)(base + db[l]); m.dec[l].wpack = (uint4*)(base + dw[l]);
  }
  const float* const* p = m.p;
  hipStream_t s = ctx->stream;
  DFA_HIP_CHECK(ctx, launch_fold_conv1(p[0], p[1], p[2], p[3], p[4], p[5], m.w1, m.b1, 32, s));
  for (int l = 0; l < 2; ++l) {
    const float* const* q = p + 6 * (l + 1);
    DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3(q[0], q[1], q[2], q[3], q[4], q[5], ecin[l], 0, ecin[l], ecout[l], precision, m.enc[l].wpack, m.enc[l].bias, s, 1, 0.25f));
  }
  {
    const float* const* q = p + 18;
    if (precision == DFA_PREC_BF16) {
      DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3(q[0], q[1], q[2], q[3], q[4], q[5], 128, 0, 128, 256, precision, m.enc[2].wpack, m.enc[2].bias, s, 1, 0.25f));
    } else {  // two Cin halves, see conv3x3_inst_cae.hip
      DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3(q[0], q[1], q[2], q[3], q[4], q[5], 128, 0, 64, 256, precision, m.enc[2].wpack, m.enc[2].bias, s, 1, 0.25f));
      DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3(q[0], q[1], q[2], q[3], q[4], q[5], 128, 64, 64, 256, precision,
                                                  m.enc[2].wpack + (size_t)(256 / 32) * 9 * 8 * 64, m.enc[2].bias, s, 1, 0.25f));
    }
  }
  for (int l = 0; l < 3; ++l) {
    const float* const* q = p + 24 + 6 * l;
    DFA_HIP_CHECK(ctx, launch_fold_pack_convt2x2(q[0], q[1], q[2], q[3], q[4], q[5], dcin[l], dcout[l], precision, m.dec[l].wpack, m.dec[l].bias, s));
  }
  m.opad_cst = (float*)(base + cst_off);
  m.c1pack = (uint4*)(base + c1p_off);
  m.c1bias = (float*)(base + c1p_off + 6 * 64 * 16);
  DFA_HIP_CHECK(ctx, launch_pack_cae_enc1_mfma(m.w1, m.b1, m.c1pack, m.c1bias, s));
  m.dec4pack = (uint4*)(base + d4p_off);
  DFA_HIP_CHECK(ctx, launch_pack_cae_dec4(p[42], m.dec4pack, s));
  if (precision == DFA_PREC_BF16)   // the constants the fused decoder uses for the columns grown from block 2's output_padding column
    DFA_HIP_CHECK(ctx, launch_cae_opad_consts(m.dec[1].bias, m.dec[2].wpack, m.dec[2].bias, p[42], p[43], m.opad_cst, s));
  m.prepared_prec = precision;
  return DFA_OK;
}

int dfa_cae_forward(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b, int64_t stride_t,
                    int64_t stride_f, const float* mu, const float* sigma, float* recon, float* latent, float* mse,
                    void* workspace, size_t workspace_bytes) {
  TraceRange trace_("dfa_cae_forward");
  if (!ctx) return DFA_E_NULL_PTR;
  CaeState& m = ctx->cae;
  if (m.prepared_prec < 0) return fail(ctx, DFA_E_NOT_PREPARED, "dfa_cae_prepare has not been called since the last set_params");
  if (!x || !workspace) return fail(ctx, DFA_E_NULL_PTR, "x and workspace must be non-null");
  if ((mu == nullptr) != (sigma == nullptr)) return fail(ctx, DFA_E_NULL_PTR, "mu and sigma must both be given or both be NULL");
  if (x_dtype != DFA_DTYPE_F32 && x_dtype != DFA_DTYPE_BF16) return fail(ctx, DFA_E_BAD_DTYPE, "x dtype %d not supported", x_dtype);
  if (B < 1) return fail(ctx, DFA_E_BAD_SHAPE, "batch must be >= 1 (got %d)", B);
  if (T < 16) return fail(ctx, DFA_E_BAD_SHAPE, "T=%d is too short: four 2x2 average pools need T >= 16", T);
  const int prec = m.prepared_prec;
  const CaePlan pl = plan_cae(B, T, F, prec);
  if (!pl.ok)
    return fail(ctx, DFA_E_BAD_SHAPE, "F=%d: decoder would rebuild %d columns (needs F = 16*(F/16)+4, e.g. 180; src/model_cae.py:68-69)", F, pl.Wd[3]);
  if (workspace_bytes < pl.total) return fail(ctx, DFA_E_WORKSPACE, "workspace too small: %zu < %zu bytes", workspace_bytes, pl.total);
  if (((uintptr_t)workspace & 255) != 0) return fail(ctx, DFA_E_WORKSPACE, "workspace must be 256-byte aligned");
  char* ws = (char*)workspace;
  void* e[4] = {ws + pl.e_off[0], ws + pl.e_off[1], ws + pl.e_off[2], ws + pl.e_off[3]};
  void* d[3] = {ws + pl.d_off[0], ws + pl.d_off[1], ws + pl.d_off[2]};
  hipStream_t s = ctx->stream;
  { ScopedSlot ts(ctx, 8);
    if (prec == DFA_PREC_BF16 && ctx->cae_enc1_mfma && F <= 1022)    // block 1 on the matrix cores (hi + lo bf16 operands)
      DFA_HIP_CHECK(ctx, launch_cae_enc1_mfma(x, x_dtype, stride_b, stride_t, stride_f, mu, sigma, m.c1pack, m.c1bias, e[0], B, T, F, s));
    else
      DFA_HIP_CHECK(ctx, launch_cae_enc1(x, x_dtype, stride_b, stride_t, stride_f, mu, sigma, m.w1, m.b1, e[0], prec, B, T, F, s)); }
  const int ecout[3] = {64, 128, 256};
  for (int l = 0; l < 3; ++l) {
    ScopedSlot ts(ctx, 9 + l);
    ConvArgs a{};
    a.in = e[l]; a.wpack = m.enc[l].wpack; a.bias = m.enc[l].bias; a.out = e[l + 1];
    a.B = B; a.H = pl.H[l + 1]; a.W = pl.W[l + 1]; a.COUT = ecout[l]; a.relu = 1; a.zero_page = ctx->zero_page;
    const int dma = ctx->cae_enc_dma;
    hipError_t err = (l == 0) ? launch_cae_enc2(prec, a, s, ctx->lds_pipe, dma) : (l == 1) ? launch_cae_enc3(prec, a, s, ctx->lds_pipe, dma)
                                                                             : launch_cae_enc4(prec, a, (float*)(ws + pl.raw_off), s, dma);
    DFA_HIP_CHECK(ctx, err);
  }
  if (latent) DFA_HIP_CHECK(ctx, launch_cae_latent_export(e[3], prec, latent, B, pl.H[4] * pl.W[4], 256, s));
  if (prec == DFA_PREC_BF16 && ctx->cae_dec_fused && cae_dec_fused_supports(T, F, stride_t, stride_f)) {   // decoder + squared error in one kernel: d1, d2, d3 never leave the CU (slot 12)
    if (recon || mse) {
      ScopedSlot ts(ctx, 12);
      float* partial = (float*)(ws + pl.part_off);
      DFA_HIP_CHECK(ctx, launch_cae_dec_fused(e[3], m.dec[0].wpack, m.dec[0].bias, m.dec[1].wpack, m.dec[1].bias, m.dec[2].wpack, m.dec[2].bias,
                                              m.p[42], m.p[43], m.dec4pack, m.opad_cst, x, x_dtype, stride_b, stride_t, stride_f, mu, sigma, recon, partial,
                                              B, pl.H[4], pl.W[4], T, F, s, ctx->clock_probe ? ctx->clock_buf : nullptr));
      if (mse) DFA_HIP_CHECK(ctx, launch_cae_mse_finalize(partial, cae_dec_fused_tiles(pl.H[4], pl.W[4]), 1.0f / ((float)T * (float)F), mse, B, s));
    }
    return DFA_OK;
  }
  const int dcin[3] = {256, 128, 64}, dcout[3] = {128, 64, 32};
  for (int l = 0; l < 3; ++l) {
    ScopedSlot ts(ctx, 12 + l);
    ConvTArgs a{};
    a.in = (l == 0) ? e[3] : d[l - 1];
    a.wpack = m.dec[l].wpack; a.bias = m.dec[l].bias; a.out = d[l];
    a.B = B; a.H = (l == 0) ? pl.H[4] : pl.Hd[l - 1]; a.W = (l == 0) ? pl.W[4] : pl.Wd[l - 1];
    a.COUT = dcout[l]; a.opad_w = (l == 1) ? 1 : 0;
    DFA_HIP_CHECK(ctx, launch_cae_dec(prec, dcin[l], a, s));
    if (l == 1) DFA_HIP_CHECK(ctx, launch_cae_opad_col(d[1], m.dec[1].bias, prec, B * pl.Hd[1], pl.Wd[1], 64, s));
  }
  if (recon || mse) {
    ScopedSlot ts(ctx, 15);
    DFA_HIP_CHECK(ctx, launch_cae_dec4_mse(d[2], prec, m.p[42], m.p[43], x, x_dtype, stride_b, stride_t, stride_f, mu, sigma,
                                           recon, (float*)(ws + pl.part_off), mse, B, pl.Hd[2], pl.Wd[2], T, F, s));
  }
  return DFA_OK;
}

}  // extern "C"
