// train_api.hip -- C ABI of the CNN2D training step (replaces, for src/train.py:71-76, what torch autograd does
// with src/model.py:13-39 in train mode):
//   dfa_cnn2d_forward_train : conv -> BatchNorm(batch statistics, running-stat update) -> ReLU -> AvgPool -> Dropout x2,
//                             conv -> BN -> ReLU -> mean_T -> Linear, keeping what backward needs in the workspace
//   dfa_cnn2d_backward      : gradients of all 14 parameters from dlogits (in parameters() order)
//   dfa_bce_smooth_fwd_bwd  : BCEWithLogitsLoss(mean) on smoothed labels + dlogits      (src/train.py:311-320)
//   dfa_adamw_step          : torch.optim.AdamW update of one flat fp32 buffer           (src/train.py:326-328)
#include "dfa_internal.h"
#include "trace.h"

using namespace dfa;

namespace dfa {
hipError_t launch_train_fwd2(int prec, const ConvArgs& a, hipStream_t s);
hipError_t launch_train_fwd3(int prec, const ConvArgs& a, hipStream_t s);
hipError_t launch_train_dgrad3(int prec, const ConvArgs& a, float* raw_tmp, hipStream_t s);
hipError_t launch_train_dgrad2(int prec, const ConvArgs& a, hipStream_t s);
enum { C1M_STATS = 0, C1M_BWD_REDUCE = 1, C1M_WGRAD = 2, C1M_BWD_FUSED = 3, C1M_STATS_XX = 4 };
enum { SRC_MEANT = 0, SRC_POOL = 1 };

// dgamma = S2, dbeta = S1 from sums[C][2]
__global__ void split_sums_kernel(const float* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                  int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { dbeta[c] = sums[2 * c]; dgamma[c] = sums[2 * c + 1]; }
}
// conv1 weight-gradient record [32][10] -> dW1[32][9], db1[32]
__global__ void split_c1_kernel(const float* __restrict__ rec, float* __restrict__ dw, float* __restrict__ db) {
  const int i = threadIdx.x;  // 320 threads
  const int c = i / 10, j = i - c * 10;
  if (j < 9) dw[c * 9 + j] = rec[i]; else db[c] = rec[i];
}
}  // namespace dfa

namespace {

inline size_t al(size_t v) { return (v + 255) / 256 * 256; }
constexpr int kWgradWGs = 256;

struct TrainPlan {
  int H1, H2;
  size_t a1, z2, a2, z3, emb, demb, msum, dz3, da2, dz2, da1, raw, stats, sums, partial, partial_bytes, total;
};

TrainPlan plan_train(int B, int T, int F, int prec) {
  TrainPlan p;
  const size_t es = (prec == DFA_PREC_BF16) ? 2 : 4;
  p.H1 = T / 2;
  p.H2 = p.H1 / 2;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = al(off + bytes); return o; };
  p.a1 = take((size_t)B * p.H1 * F * 32 * es);
  p.z2 = take((size_t)B * p.H1 * F * 64 * es);
  p.a2 = take((size_t)B * p.H2 * F * 64 * es);
  p.z3 = take((size_t)B * p.H2 * F * 128 * es);
  p.emb = take((size_t)B * 128 * F * 4);
  p.demb = take((size_t)B * 128 * F * 4);
  p.msum = take((size_t)B * 128 * F * 2 * 4);        // block 3: per (b, f, c) mask count and mask*xhat sum over t (bn_relu_meant)
  p.dz3 = take((size_t)B * p.H2 * F * 128 * es);
  p.da2 = take((size_t)B * p.H2 * F * 64 * es);
  p.dz2 = take((size_t)B * p.H1 * F * 64 * es);
  p.da1 = take((size_t)B * p.H1 * F * 32 * es);
  p.raw = take((size_t)B * p.H2 * F * 64 * 4);
  p.stats = take((32 + 64 + 128) * 3 * 4);          // mean | var | invstd per layer
  p.sums = take((32 + 64 + 128) * 2 * 4 + 352 * 4 + 96 * 4);  // (S1,S2) per layer + conv1 backward record [32][11] + XX[9][9] | Xs[9]
  const int nstrips = (F + 29) / 30;                                         // (the 30-column strips of conv3_m16: >= the 32-column count)
  size_t pb = (size_t)B * nstrips * 128 * 2 * 4;                              // conv stats partials
  pb = std::max(pb, ((size_t)conv1_train_blocks(B, T, F) + 64) * 352 * 4);   // conv1 passes + 2nd-level scratch
  int ppb;
  pb = std::max(pb, ((size_t)bn_bwd_blocks(B, p.H1, F, &ppb) + 64) * 128 * 2 * 4);  // BN backward partials + 2nd-level scratch
  pb = std::max(pb, (size_t)kWgradWGs * ((size_t)128 * 64 * 9 + 256) * 4);   // weight-gradient partials
  pb = std::max(pb, ((size_t)B * F / 128 + 1) * 128 * 2 * 4);                // saved-sum BN reduction partials (8 * 16 positions per block)
  p.partial_bytes = pb;
  p.partial = take(pb);
  p.total = off;
  return p;
}

struct StatPtrs { float *mean, *var, *invstd; };
// BatchNorm batch statistics from the per-workgroup records partial[nparts][C][2].  Synchronised BatchNorm (dfa_ctx_set_bn_sync):
// the records are first reduced to one [C][2] record in the caller's buffer, summed over the ranks by the caller's hook, and the
// statistics come from those sums and the global count -- every rank ends with the same mean / variance / running statistics.
static int finalize_bn_stats(dfa_ctx* ctx, const float* partial, int nparts, int C, double n, float* mean, float* var, float* invstd,
                             float* rm, float* rv, float momentum, float* scratch) {
  const dfa::BnSync& sy = ctx->bn_sync;
  hipStream_t s = ctx->stream;
  if (!sy.fn) {
    DFA_HIP_CHECK(ctx, launch_bn_finalize(partial, nparts, C, n, mean, var, invstd, rm, rv, momentum, s));
    return DFA_OK;
  }
  DFA_HIP_CHECK(ctx, launch_reduce_partials(partial, nparts, C * 2, 1.0f, sy.buf, s, scratch));
  if (sy.fn(sy.user, sy.buf, C * 2) != 0) return fail(ctx, DFA_E_HIP, "the BatchNorm synchronisation hook failed (forward statistics, %d channels)", C);
  DFA_HIP_CHECK(ctx, launch_bn_finalize(sy.buf, 1, C, n * (double)sy.world, mean, var, invstd, rm, rv, momentum, s));
  return DFA_OK;
}
StatPtrs stat_ptrs(char* ws, const TrainPlan& pl, int layer) {
  const int off[3] = {0, 32, 96}, C[3] = {32, 64, 128};
  float* base = (float*)(ws + pl.stats);
  float* l = base + 3 * off[layer];
  return {l, l + C[layer], l + 2 * C[layer]};
}
float* sums_ptr(char* ws, const TrainPlan& pl, int layer) {
  const int off[3] = {0, 32, 96};
  return (float*)(ws + pl.sums) + 2 * off[layer];
}

}  // namespace

extern "C" {

size_t dfa_cnn2d_train_workspace_bytes(const dfa_ctx* ctx, int B, int T, int F, int precision) {
  (void)ctx;
  if (B < 1 || T < 4 || F < 1) return 0;
  return plan_train(B, T, F, precision).total;
}

int dfa_cnn2d_forward_train(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b,
                            int64_t stride_t, int64_t stride_f, int precision, float p_drop, uint64_t seed,
                            uint64_t offset, float momentum, int update_running_stats, float* logits, float* embedding,
                            void* workspace, size_t workspace_bytes) {
  TraceRange trace_("dfa_cnn2d_forward_train");
  if (!ctx) return DFA_E_NULL_PTR;
  Cnn2dState& m = ctx->cnn2d;
  // The augmentation armed by dfa_cnn2d_set_train_augment is one-shot and belongs to THIS call: it is taken (and the arm cleared)
  // before any check can return, so a failed forward never leaves it armed for an unrelated later batch.
  const AugCfg armed = m.aug_armed;
  m.aug_armed = AugCfg{};
  m.train_aug = AugCfg{};
  if (!m.have_params) return fail(ctx, DFA_E_NOT_PREPARED, "dfa_cnn2d_set_params has not been called");
  if (!x || !logits || !workspace) return fail(ctx, DFA_E_NULL_PTR, "x, logits and workspace must be non-null");
  if (x_dtype != DFA_DTYPE_F32 && x_dtype != DFA_DTYPE_BF16) return fail(ctx, DFA_E_BAD_DTYPE, "x dtype %d not supported", x_dtype);
  if (precision != DFA_PREC_F32 && precision != DFA_PREC_BF16) return fail(ctx, DFA_E_BAD_DTYPE, "unknown precision %d", precision);
  if (armed.on && (armed.T != T || armed.F != F))
    return fail(ctx, DFA_E_BAD_SHAPE, "the armed augmentation was drawn for [T=%d, F=%d], the batch is [T=%d, F=%d]", armed.T, armed.F, T, F);
  if (B < 1 || T < 4) return fail(ctx, DFA_E_BAD_SHAPE, "need B >= 1 and T >= 4 (got %d, %d)", B, T);
  if (F != m.in_features) return fail(ctx, DFA_E_BAD_SHAPE, "feature dim %d does not match in_features=%d", F, m.in_features);
  if (!(p_drop >= 0.f && p_drop < 1.f)) return fail(ctx, DFA_E_BAD_SHAPE, "dropout p must be in [0, 1)");
  const TrainPlan pl = plan_train(B, T, F, precision);
  if (workspace_bytes < pl.total) return fail(ctx, DFA_E_WORKSPACE, "train workspace too small: %zu < %zu bytes", workspace_bytes, pl.total);
  if (((uintptr_t)workspace & 255) != 0) return fail(ctx, DFA_E_WORKSPACE, "workspace must be 256-byte aligned");
  DFA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // train-mode weight images: raw convs (BN is its own pass) + data-gradient images; rebuilt every step (weights move)
  if (!m.train_packed) {
    const size_t w2 = (size_t)64 * 32 * 9 * 4, w3 = (size_t)128 * 64 * 9 * 4;
    const size_t need = al((288 + 32 + 64 + 128 + 32 + 64) * 4) + 2 * (w2 + w3);
    DFA_HIP_CHECK(ctx, hipMalloc(&m.train_packed, need));
    char* base = (char*)m.train_packed;
    m.tw1 = (float*)base; m.tb1 = m.tw1 + 288;
    m.t2.bias = m.tb1 + 32; m.t3.bias = m.t2.bias + 64;
    m.d2.bias = m.t3.bias + 128; m.d3.bias = m.d2.bias + 32;
    char* wp = base + al((288 + 32 + 64 + 128 + 32 + 64) * 4);
    m.t2.wpack = (uint4*)wp; wp += w2;
    m.t3.wpack = (uint4*)wp; wp += w3;
    m.d2.wpack = (uint4*)wp; wp += w2;
    m.d3.wpack = (uint4*)wp;
  }
  const float* const* p = m.p;
  hipStream_t s = ctx->stream;
  const int prec = precision;
  DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3(p[6], p[7], nullptr, nullptr, nullptr, nullptr, 32, 0, 32, 64, prec, m.t2.wpack, m.t2.bias, s, 0));
  DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3(p[12], p[13], nullptr, nullptr, nullptr, nullptr, 64, 0, 64, 128, prec, m.t3.wpack, m.t3.bias, s, 0));
  // bf16: the 16x16x32 image of the same raw weights, in the unused second half of the (fp32-sized) t3 buffer
  uint4* t3_m16 = (uint4*)((char*)m.t3.wpack + (size_t)128 * 64 * 9 * 2);
  const bool fwd3_m16 = prec == DFA_PREC_BF16 && train_conv_variant() != 1;   // 2: pipelined, 0: its compiler-scheduled twin
  if (fwd3_m16)
    DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3_m16(p[12], p[13], nullptr, nullptr, nullptr, nullptr, 64, 128, t3_m16, s, 0));
  m.train_dgrad_m16 = (prec == DFA_PREC_BF16 && ctx->dgrad_m16) ? 1 : 0;
  if (m.train_dgrad_m16) {
    DFA_HIP_CHECK(ctx, launch_pack_conv3x3_dgrad_m16(p[6], 32, 64, m.d2.wpack, m.d2.bias, s));
    DFA_HIP_CHECK(ctx, launch_pack_conv3x3_dgrad_m16(p[12], 64, 128, m.d3.wpack, m.d3.bias, s));
  } else {
  DFA_HIP_CHECK(ctx, launch_pack_conv3x3_dgrad(p[6], 32, 64, 0, 64, prec, m.d2.wpack, m.d2.bias, s));
  {  // two Cin halves (see launch_train_dgrad3)
    const int nkg = (prec == DFA_PREC_BF16) ? 4 : 8;
    DFA_HIP_CHECK(ctx, launch_pack_conv3x3_dgrad(p[12], 64, 128, 0, 64, prec, m.d3.wpack, m.d3.bias, s));
    DFA_HIP_CHECK(ctx, launch_pack_conv3x3_dgrad(p[12], 64, 128, 64, 64, prec, m.d3.wpack + (size_t)(64 / 32) * 9 * nkg * 64, m.d3.bias, s));
  }
  }
  char* ws = (char*)workspace;
  float* partial = (float*)(ws + pl.partial);
  DropCfg dc{};
  dc.thresh = (p_drop > 0.f) ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  dc.scale = 1.0f / (1.0f - p_drop);
  dc.seed = seed; dc.offset = offset;
  m.train_drop = dc; m.train_prec = prec; m.train_B = B; m.train_T = T;
  float* rm[3] = {nullptr, nullptr, nullptr}; float* rv[3] = {nullptr, nullptr, nullptr};
  if (update_running_stats) {
    rm[0] = (float*)p[4]; rv[0] = (float*)p[5]; rm[1] = (float*)p[10]; rv[1] = (float*)p[11];
    rm[2] = (float*)p[16]; rv[2] = (float*)p[17];
  }
  // ---- block 1
  StatPtrs s1 = stat_ptrs(ws, pl, 0);
  // augmentation armed by dfa_cnn2d_set_train_augment: one-shot, folded into the three kernels that read x
  m.train_aug = armed;
  const AugCfg* aug = m.train_aug.on ? &m.train_aug : nullptr;
  // synchronised BatchNorm needs the layer's (sum dy, sum dy*xhat) BEFORE the weight gradient is formed: block 1 then takes the
  // two-pass vector path (reduce -> hook -> weight gradient), not the one-pass moment algebra
  m.train_c1_fused = (ctx->conv1_bwd_fused && !ctx->bn_sync.fn) ? 1 : 0;
  // matrix-core passes: bf16 mode on bf16 features without a folded augmentation (its noise makes x non-bf16), fused backward
  // (the backward reads da1 with the dropout keep mask already applied by the 16x16x32 data-gradient kernel)
  m.train_c1_mfma = (ctx->conv1_mfma && m.train_c1_fused && prec == DFA_PREC_BF16 && x_dtype == DFA_DTYPE_BF16 && !aug && F <= 224 &&
                     (dc.thresh == 0 || m.train_dgrad_m16)) ? 1 : 0;
  const int nb1f = m.train_c1_mfma ? conv1_mfma_blocks(B, T, F) : conv1_train_blocks(B, T, F);
  if (m.train_c1_mfma)
    DFA_HIP_CHECK(ctx, launch_conv1_mfma(C1X_STATS, x, stride_b, stride_t, stride_f, p[0], p[1], nullptr, nullptr, partial, B, T, F, dc, s));
  else
  DFA_HIP_CHECK(ctx, launch_conv1_train(m.train_c1_fused ? C1M_STATS_XX : C1M_STATS, x, x_dtype, stride_b, stride_t, stride_f, p[0], p[1],
                                        nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, prec, partial, B, T, F, dc, s, 1, aug));
  { const int rc = finalize_bn_stats(ctx, partial, nb1f, 32, (double)B * T * F, s1.mean, s1.var, s1.invstd, rm[0], rv[0], momentum, partial + (size_t)nb1f * 352);
    if (rc != DFA_OK) return rc; }
  if (m.train_c1_fused) {   // XX[9][9] | Xs[9] (block records of 96 floats behind the [32][2] records) -> the sums region, for backward
    float* xxs = (float*)(ws + pl.sums) + 2 * (32 + 64 + 128) + 352;
    DFA_HIP_CHECK(ctx, launch_reduce_partials(partial + (size_t)nb1f * 64, nb1f, 96, 1.0f, xxs, s, partial + (size_t)nb1f * 160));
  }
  DFA_HIP_CHECK(ctx, launch_fold_conv1(p[0], p[1], p[2], p[3], s1.mean, s1.var, m.tw1, m.tb1, 32, s));
  dc.layer = 1;
  if (m.train_c1_mfma)
    DFA_HIP_CHECK(ctx, launch_conv1_mfma(C1X_FWD, x, stride_b, stride_t, stride_f, m.tw1, m.tb1, ws + pl.a1, nullptr, nullptr, B, T, F, dc, s));
  else
  DFA_HIP_CHECK(ctx, launch_conv1(x, x_dtype, stride_b, stride_t, stride_f, m.tw1, m.tb1, ws + pl.a1, prec, B, T, F, s, &dc, aug));
  // ---- block 2
  const int nstrips = (F + 31) / 32;
  {
    ConvArgs a{};
    a.in = ws + pl.a1; a.wpack = m.t2.wpack; a.bias = m.t2.bias; a.out = ws + pl.z2;
    a.B = B; a.H = pl.H1; a.W = F; a.COUT = 64; a.relu = 0; a.stats_partial = partial; a.zero_page = ctx->zero_page;
    DFA_HIP_CHECK(ctx, launch_train_fwd2(prec, a, s));
  }
  StatPtrs s2 = stat_ptrs(ws, pl, 1);
  { const int rc = finalize_bn_stats(ctx, partial, B * nstrips, 64, (double)B * pl.H1 * F, s2.mean, s2.var, s2.invstd, rm[1], rv[1], momentum, partial + (size_t)B * nstrips * 128);
    if (rc != DFA_OK) return rc; }
  dc.layer = 2;
  DFA_HIP_CHECK(ctx, launch_bn_relu_pool_drop(prec, ws + pl.z2, s2.mean, s2.invstd, p[8], p[9], ws + pl.a2, B, pl.H1, F, 64, dc, s));
  // ---- block 3
  {
    ConvArgs a{};
    a.in = ws + pl.a2; a.wpack = m.t3.wpack; a.bias = m.t3.bias; a.out = ws + pl.z3;
    a.B = B; a.H = pl.H2; a.W = F; a.COUT = 128; a.relu = 0; a.stats_partial = partial; a.zero_page = ctx->zero_page;
    if (fwd3_m16) {
      a.wpack = t3_m16;
      DFA_HIP_CHECK(ctx, launch_train_fwd3_m16(a, s, train_conv_variant() == 2));
    } else {
      DFA_HIP_CHECK(ctx, launch_train_fwd3(prec, a, s));
    }
  }
  StatPtrs s3 = stat_ptrs(ws, pl, 2);
  { const int np3 = B * (fwd3_m16 ? (F + 29) / 30 : nstrips);   // conv3_m16 owns 30 columns per strip
    const int rc = finalize_bn_stats(ctx, partial, np3, 128, (double)B * pl.H2 * F, s3.mean, s3.var, s3.invstd, rm[2], rv[2], momentum, partial + (size_t)np3 * 256);
    if (rc != DFA_OK) return rc; }
  float* emb = (float*)(ws + pl.emb);
  DFA_HIP_CHECK(ctx, launch_bn_relu_meant(prec, ws + pl.z3, s3.mean, s3.invstd, p[14], p[15], emb, B, pl.H2, F, 128, s, (float*)(ws + pl.msum)));
  if (embedding) DFA_HIP_CHECK(ctx, hipMemcpyAsync(embedding, emb, (size_t)B * 128 * F * 4, hipMemcpyDeviceToDevice, s));
  DFA_HIP_CHECK(ctx, launch_linear(emb, p[18], p[19], logits, B, 128 * F, s));
  return DFA_OK;
}

int dfa_cnn2d_backward(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b, int64_t stride_t,
                       int64_t stride_f, const float* dlogits, float* const* grads, int ngrads, void* workspace,
                       size_t workspace_bytes) {
  TraceRange trace_("dfa_cnn2d_backward");
  if (!ctx) return DFA_E_NULL_PTR;
  Cnn2dState& m = ctx->cnn2d;
  if (!m.train_packed || m.train_B != B || m.train_T != T)
    return fail(ctx, DFA_E_NOT_PREPARED, "dfa_cnn2d_backward must follow dfa_cnn2d_forward_train on the same batch");
  if (!x || !dlogits || !grads || !workspace) return fail(ctx, DFA_E_NULL_PTR, "x, dlogits, grads and workspace must be non-null");
  if (ngrads != 14) return fail(ctx, DFA_E_BAD_SHAPE, "cnn2d has 14 parameters, got %d gradient pointers", ngrads);
  for (int i = 0; i < 14; ++i)
    if (!grads[i]) return fail(ctx, DFA_E_NULL_PTR, "gradient pointer %d is null", i);
  const int prec = m.train_prec;
  const TrainPlan pl = plan_train(B, T, F, prec);
  if (workspace_bytes < pl.total) return fail(ctx, DFA_E_WORKSPACE, "train workspace too small");
  char* ws = (char*)workspace;
  float* partial = (float*)(ws + pl.partial);
  const float* const* p = m.p;
  hipStream_t s = ctx->stream;
  DropCfg dc = m.train_drop;
  StatPtrs s1 = stat_ptrs(ws, pl, 0), s2 = stat_ptrs(ws, pl, 1), s3 = stat_ptrs(ws, pl, 2);
  float *sm1 = sums_ptr(ws, pl, 0), *sm2 = sums_ptr(ws, pl, 1), *sm3 = sums_ptr(ws, pl, 2);
  float* c1rec = (float*)(ws + pl.sums) + 2 * (32 + 64 + 128);
  float* demb = (float*)(ws + pl.demb);
  const dfa::BnSync* sync = ctx->bn_sync.fn ? &ctx->bn_sync : nullptr;     // synchronised BatchNorm (dfa_ctx_set_bn_sync)
  // classifier
  DFA_HIP_CHECK(ctx, launch_linear_bwd(dlogits, p[18], (const float*)(ws + pl.emb), demb, grads[12], grads[13], B, 128 * F, s, 128, F));
  // block 3: BN backward (upstream = mean_T then Linear), weight gradient, data gradient
  DFA_HIP_CHECK(ctx, launch_bn_bwd_meant_saved(prec, ws + pl.z3, s3.mean, s3.invstd, p[14], p[15], demb, (const float*)(ws + pl.msum), partial,
                                               sm3, ws + pl.dz3, B, pl.H2, F, 128, s, sync));
  hipLaunchKernelGGL(split_sums_kernel, dim3(1), dim3(128), 0, s, sm3, grads[10], grads[11], 128);
  DFA_HIP_CHECK(ctx, launch_wgrad3x3(prec, 64, 128, ws + pl.dz3, ws + pl.a2, partial, grads[8], grads[9], B, pl.H2, F, kWgradWGs, s));
  {
    ConvArgs a{};
    a.in = ws + pl.dz3; a.wpack = m.d3.wpack; a.bias = m.d3.bias; a.out = ws + pl.da2;
    a.B = B; a.H = pl.H2; a.W = F; a.COUT = 64; a.relu = 0; a.zero_page = ctx->zero_page;
    if (m.train_dgrad_m16) DFA_HIP_CHECK(ctx, launch_train_dgrad3_m16(a, s, train_conv_variant() != 0));
    else DFA_HIP_CHECK(ctx, launch_train_dgrad3(prec, a, (float*)(ws + pl.raw), s));
  }
  // block 2
  dc.layer = 2;
  {
    int ppb_;
    float* scratch2 = partial + (size_t)bn_bwd_blocks(B, pl.H1, F, &ppb_) * 64 * 2;
    DFA_HIP_CHECK(ctx, launch_bn_bwd(prec, SRC_POOL, ws + pl.z2, s2.mean, s2.invstd, p[8], p[9], nullptr, ws + pl.da2, partial, sm2, ws + pl.dz2,
                                     B, pl.H1, F, 64, dc, s, scratch2, sync));
  }
  hipLaunchKernelGGL(split_sums_kernel, dim3(1), dim3(128), 0, s, sm2, grads[6], grads[7], 64);
  DFA_HIP_CHECK(ctx, launch_wgrad3x3(prec, 32, 64, ws + pl.dz2, ws + pl.a1, partial, grads[4], grads[5], B, pl.H1, F, kWgradWGs, s));
  {
    ConvArgs a{};
    a.in = ws + pl.dz2; a.wpack = m.d2.wpack; a.bias = m.d2.bias; a.out = ws + pl.da1;
    a.B = B; a.H = pl.H1; a.W = F; a.COUT = 32; a.relu = 0; a.zero_page = ctx->zero_page;
    if (m.train_c1_mfma) { a.drop = dc; a.drop.layer = 1; }   // keep mask of a1's dropout where da1 is produced (idempotent for the vector kernel)
    if (m.train_dgrad_m16) DFA_HIP_CHECK(ctx, launch_train_dgrad2_m16(a, s, train_conv_variant() != 0));
    else DFA_HIP_CHECK(ctx, launch_train_dgrad2(prec, a, s));
  }
  // block 1 (z1 recomputed from x)
  dc.layer = 1;
  const int nb1 = conv1_train_blocks(B, T, F);
  const AugCfg* aug = m.train_aug.on ? &m.train_aug : nullptr;
  if (m.train_c1_mfma && ctx->conv1_mfma) {   // (the option may be cleared between forward and backward: the vector kernel reads the same forward state -- twin test)
    // the ReLU mask comes from the forward's own folded image (m.tw1 / m.tb1 are this step's); S2 is derived in the finalize
    const int nbm = conv1_mfma_blocks(B, T, F);
    DFA_HIP_CHECK(ctx, launch_conv1_mfma(C1X_BWD, x, stride_b, stride_t, stride_f, m.tw1, m.tb1, nullptr, ws + pl.da1, partial, B, T, F, dc, s));
    DFA_HIP_CHECK(ctx, launch_reduce_partials(partial, nbm, 352, 1.0f, c1rec, s, partial + (size_t)nbm * 352));
    DFA_HIP_CHECK(ctx, launch_conv1_bwd_finalize(c1rec, c1rec + 352, p[0], p[1], s1.mean, s1.invstd, p[2], (double)B * T * F, grads[0], grads[1],
                                                 grads[2], grads[3], s, 1));
    DFA_HIP_CHECK(ctx, hipGetLastError());
    return DFA_OK;
  }
  if (m.train_c1_fused) {
    DFA_HIP_CHECK(ctx, launch_conv1_train(C1M_BWD_FUSED, x, x_dtype, stride_b, stride_t, stride_f, p[0], p[1], s1.mean, s1.invstd, p[2], p[3],
                                          nullptr, ws + pl.da1, prec, partial, B, T, F, dc, s, 1, aug));
    float* rec = c1rec;                       // [32][11]
    DFA_HIP_CHECK(ctx, launch_reduce_partials(partial, nb1, 352, 1.0f, rec, s, partial + (size_t)nb1 * 352));
    DFA_HIP_CHECK(ctx, launch_conv1_bwd_finalize(rec, c1rec + 352, p[0], p[1], s1.mean, s1.invstd, p[2], (double)B * T * F, grads[0], grads[1],
                                                 grads[2], grads[3], s));
    DFA_HIP_CHECK(ctx, hipGetLastError());
    return DFA_OK;
  }
  DFA_HIP_CHECK(ctx, launch_conv1_train(C1M_BWD_REDUCE, x, x_dtype, stride_b, stride_t, stride_f, p[0], p[1], s1.mean, s1.invstd, p[2], p[3],
                                        nullptr, ws + pl.da1, prec, partial, B, T, F, dc, s, 1, aug));
  float* scratch = partial + (size_t)nb1 * 320;
  DFA_HIP_CHECK(ctx, launch_reduce_partials(partial, nb1, 64, 1.0f, sm1, s, scratch));
  hipLaunchKernelGGL(split_sums_kernel, dim3(1), dim3(128), 0, s, sm1, grads[2], grads[3], 32);      // dgamma, dbeta: this rank's own sums
  const float* sm1_a;
  float isc1;
  DFA_HIP_CHECK(ctx, bn_sync_sums(sync, sm1, 64, s, &sm1_a, &isc1));                                   // dz1 is formed from the global ones
  DFA_HIP_CHECK(ctx, launch_conv1_train(C1M_WGRAD, x, x_dtype, stride_b, stride_t, stride_f, p[0], p[1], s1.mean, s1.invstd, p[2], p[3],
                                        sm1_a, ws + pl.da1, prec, partial, B, T, F, dc, s, 1, aug, isc1));
  DFA_HIP_CHECK(ctx, launch_reduce_partials(partial, nb1, 320, 1.0f, c1rec, s, scratch));
  hipLaunchKernelGGL(split_c1_kernel, dim3(1), dim3(320), 0, s, c1rec, grads[0], grads[1]);
  DFA_HIP_CHECK(ctx, hipGetLastError());
  return DFA_OK;
}

// shared by the CNN2D and CNN1D entry points: validate, copy the keep mask into the context's double buffer, fill `armed`
static int arm_train_augment(dfa_ctx* ctx, AugCfg& armed, int enable, int T, int F, int shift, const float* keep_f, int tmask_start,
                             int tmask_len, int fmask_start, int fmask_len, float jitter_std, uint64_t seed, uint64_t offset) {
  armed = AugCfg{};
  if (!enable) return DFA_OK;
  if (T < 1 || F < 1) return fail(ctx, DFA_E_BAD_SHAPE, "bad augmentation shape [T=%d, F=%d]", T, F);
  if (tmask_len < 0 || fmask_len < 0 || tmask_start < 0 || fmask_start < 0 || tmask_start + tmask_len > T ||
      fmask_start + fmask_len > F)
    return fail(ctx, DFA_E_BAD_SHAPE, "mask span outside the batch");
  if (!(jitter_std >= 0.f)) return fail(ctx, DFA_E_BAD_SHAPE, "jitter std must be >= 0");
  // The keep mask is COPIED (stream-ordered) into one of two context-owned buffers, alternating per call: the caller's tensor may
  // be freed as soon as this returns (a temporary FusedAugment did exactly that in a test and the backward read recycled memory),
  // and arming batch n+1 cannot disturb the backward of batch n.
  const float* keep_dev = nullptr;
  if (keep_f) {
    DFA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (ctx->aug_keep_cap < F) {
      if (ctx->aug_keep) { DFA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); DFA_HIP_CHECK(ctx, hipFree(ctx->aug_keep)); ctx->aug_keep = nullptr; }
      const int cap = F < 1024 ? 1024 : F;
      DFA_HIP_CHECK(ctx, hipMalloc((void**)&ctx->aug_keep, (size_t)2 * cap * sizeof(float)));
      ctx->aug_keep_cap = cap;
    }
    ctx->aug_keep_slot ^= 1;
    float* dst = ctx->aug_keep + (size_t)ctx->aug_keep_slot * ctx->aug_keep_cap;
    DFA_HIP_CHECK(ctx, hipMemcpyAsync(dst, keep_f, (size_t)F * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    keep_dev = dst;
  }
  AugCfg a{};
  a.on = 1; a.T = T; a.F = F; a.shift = ((shift % T) + T) % T; a.keep = keep_dev;
  a.tm_start = tmask_start; a.tm_len = tmask_len; a.fm_start = fmask_start; a.fm_len = fmask_len;
  a.std = jitter_std; a.seed = seed; a.offset = offset;
  armed = a;
  return DFA_OK;
}

int dfa_cnn2d_set_train_augment(dfa_ctx* ctx, int enable, int T, int F, int shift, const float* keep_f, int tmask_start,
                                int tmask_len, int fmask_start, int fmask_len, float jitter_std, uint64_t seed,
                                uint64_t offset) {
  if (!ctx) return DFA_E_NULL_PTR;
  return arm_train_augment(ctx, ctx->cnn2d.aug_armed, enable, T, F, shift, keep_f, tmask_start, tmask_len, fmask_start, fmask_len,
                           jitter_std, seed, offset);
}

int dfa_cnn1d_set_train_augment(dfa_ctx* ctx, int enable, int T, int F, int shift, const float* keep_f, int tmask_start,
                                int tmask_len, int fmask_start, int fmask_len, float jitter_std, uint64_t seed,
                                uint64_t offset) {
  if (!ctx) return DFA_E_NULL_PTR;
  return arm_train_augment(ctx, ctx->cnn1d.aug_armed, enable, T, F, shift, keep_f, tmask_start, tmask_len, fmask_start, fmask_len,
                           jitter_std, seed, offset);
}

int dfa_ctx_set_bn_sync(dfa_ctx* ctx, dfa_bn_sync_fn fn, void* user, int world, float* buf, int capacity) {
  if (!ctx) return DFA_E_NULL_PTR;
  ctx->bn_sync = dfa::BnSync{};
  if (!fn) return DFA_OK;
  if (world < 1) return fail(ctx, DFA_E_BAD_SHAPE, "world must be >= 1 (got %d)", world);
  if (!buf || capacity < 512) return fail(ctx, DFA_E_NULL_PTR, "the synchronisation buffer must hold at least 512 floats");
  ctx->bn_sync.fn = fn; ctx->bn_sync.user = user; ctx->bn_sync.world = world; ctx->bn_sync.buf = buf;
  return DFA_OK;
}

int dfa_bce_smooth_fwd_bwd(dfa_ctx* ctx, const float* logits, const float* labels, float label_smoothing, int B,
                           float* loss, float* dlogits) {
  if (!ctx) return DFA_E_NULL_PTR;
  if (!logits || !labels) return fail(ctx, DFA_E_NULL_PTR, "logits and labels must be non-null");
  if (!(label_smoothing >= 0.f && label_smoothing < 0.5f))
    return fail(ctx, DFA_E_BAD_SHAPE, "--label-smoothing must be in [0, 0.5)");   /* src/train.py:308-309 */
  if (B < 1) return fail(ctx, DFA_E_BAD_SHAPE, "B must be >= 1");
  DFA_HIP_CHECK(ctx, launch_bce_smooth(logits, labels, label_smoothing, B, loss, dlogits, ctx->stream));
  return DFA_OK;
}

int dfa_adamw_step(dfa_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale) {
  TraceRange trace_("dfa_adamw_step");
  if (!ctx) return DFA_E_NULL_PTR;
  if (!param || !grad || !exp_avg || !exp_avg_sq) return fail(ctx, DFA_E_NULL_PTR, "adamw buffers must be non-null");
  if (step < 1) return fail(ctx, DFA_E_BAD_SHAPE, "step is 1-based (got %d)", step);
  if (n == 0) return DFA_OK;
  DFA_HIP_CHECK(ctx, launch_adamw(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, ctx->stream));
  return DFA_OK;
}

}  // extern "C"

/* ------------------------------------------------------------------------------------------------ CNN1D training */
namespace {

struct Train1dPlan {
  size_t z[3], h[2], pooled, dpooled, dz[3], dh[2], stats, sums, partial, total;
};

Train1dPlan plan_train1d(int B, int T, int F) {
  Train1dPlan p;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = al(off + bytes); return o; };
  const int C[3] = {32, 64, 128};
  for (int l = 0; l < 3; ++l) p.z[l] = take((size_t)B * C[l] * T * 4);
  for (int l = 0; l < 2; ++l) p.h[l] = take((size_t)B * C[l] * T * 4);
  p.pooled = take((size_t)B * 128 * 4);
  p.dpooled = take((size_t)B * 128 * 4);
  for (int l = 0; l < 3; ++l) p.dz[l] = take((size_t)B * C[l] * T * 4);
  for (int l = 0; l < 2; ++l) p.dh[l] = take((size_t)B * C[l] * T * 4);
  p.stats = take((32 + 64 + 128) * 3 * 4);
  p.sums = take((32 + 64 + 128) * 2 * 4);
  const size_t nch = (size_t)cm_chunks(B);
  size_t pb = nch * 128 * 2 * 4;
  const size_t wch = (size_t)conv1d_wgrad_chunks(B);
  pb = std::max(pb, wch * ((size_t)32 * F * 3 + 32) * 4);
  pb = std::max(pb, wch * ((size_t)128 * 64 * 3 + 128) * 4);
  p.partial = take(pb);
  p.total = off;
  return p;
}

struct St { float *mean, *var, *invstd; };
St st1d(char* ws, const Train1dPlan& pl, int l) {
  const int off[3] = {0, 32, 96}, C[3] = {32, 64, 128};
  float* b = (float*)(ws + pl.stats) + 3 * off[l];
  return {b, b + C[l], b + 2 * C[l]};
}
float* sums1d(char* ws, const Train1dPlan& pl, int l) {
  const int off[3] = {0, 32, 96};
  return (float*)(ws + pl.sums) + 2 * off[l];
}

}  // namespace

extern "C" {

size_t dfa_cnn1d_train_workspace_bytes(const dfa_ctx* ctx, int B, int T, int F) {
  (void)ctx;
  if (B < 1 || T < 1 || F < 1) return 0;
  return plan_train1d(B, T, F).total;
}

int dfa_cnn1d_forward_train(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b,
                            int64_t stride_t, int64_t stride_f, float p_drop, uint64_t seed, uint64_t offset,
                            float momentum, int update_running_stats, float* logits, void* workspace,
                            size_t workspace_bytes) {
  TraceRange trace_("dfa_cnn1d_forward_train");
  if (!ctx) return DFA_E_NULL_PTR;
  Cnn1dState& m = ctx->cnn1d;
  const AugCfg armed = m.aug_armed;     // one-shot: consumed here, also by a call that fails its checks below
  m.aug_armed = AugCfg{};
  m.train_aug = AugCfg{};
  if (!m.have_params) return fail(ctx, DFA_E_NOT_PREPARED, "dfa_cnn1d_set_params has not been called");
  if (!x || !logits || !workspace) return fail(ctx, DFA_E_NULL_PTR, "x, logits and workspace must be non-null");
  if (x_dtype != DFA_DTYPE_F32) return fail(ctx, DFA_E_BAD_DTYPE, "cnn1d takes float32 input (got dtype %d)", x_dtype);
  if (B < 1 || T < 1) return fail(ctx, DFA_E_BAD_SHAPE, "B and T must be >= 1 (got %d, %d)", B, T);
  if (F != m.in_features) return fail(ctx, DFA_E_BAD_SHAPE, "feature dim %d does not match in_features=%d", F, m.in_features);
  if (!(p_drop >= 0.f && p_drop < 1.f)) return fail(ctx, DFA_E_BAD_SHAPE, "dropout p must be in [0, 1)");
  const Train1dPlan pl = plan_train1d(B, T, F);
  if (workspace_bytes < pl.total) return fail(ctx, DFA_E_WORKSPACE, "train workspace too small: %zu < %zu bytes", workspace_bytes, pl.total);
  if (armed.on && (armed.T != T || armed.F != F))
    return fail(ctx, DFA_E_BAD_SHAPE, "armed augmentation is for [T=%d, F=%d], the batch is [T=%d, F=%d]", armed.T, armed.F, T, F);
  m.train_aug = armed;
  const AugCfg* aug = m.train_aug.on ? &m.train_aug : nullptr;
  DFA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!m.train_packed) {
    const size_t n = al((size_t)64 * 32 * 3 * 4) + al((size_t)128 * 64 * 3 * 4) + al(256 * 4);
    DFA_HIP_CHECK(ctx, hipMalloc(&m.train_packed, n));
    char* b = (char*)m.train_packed;
    m.wt[0] = (float*)b;
    m.wt[1] = (float*)(b + al((size_t)64 * 32 * 3 * 4));
    m.zero_bias = (float*)(b + al((size_t)64 * 32 * 3 * 4) + al((size_t)128 * 64 * 3 * 4));
  }
  // bf16x3 A-fragment images of this step's weights (they change every step): forward layers 1-3, data gradients 3->2, 2->1
  const int xcin[5] = {F, 32, 64, 128, 64}, xcout[5] = {32, 64, 128, 64, 32};
  if (!m.wx3[0] || m.wx3_F != F) {
    if (m.wx3[0]) { DFA_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); DFA_HIP_CHECK(ctx, hipFree(m.wx3[0])); m.wx3[0] = nullptr; }
    size_t off[6] = {0};
    for (int i = 0; i < 5; ++i) off[i + 1] = off[i] + al(conv1d_terms_pack_bytes(xcin[i], xcout[i], 3));
    char* base = nullptr;
    DFA_HIP_CHECK(ctx, hipMalloc((void**)&base, off[5]));
    for (int i = 0; i < 5; ++i) m.wx3[i] = base + off[i];
    m.wx3_F = F;
  }
  const float* const* p = m.p;
  hipStream_t s = ctx->stream;
  const int x3 = ctx->cnn1d_train_x3, terms = (x3 == 3) ? 2 : 3;
  // the fp32 data-gradient images are only read by the vector-ALU fallback (T > 384, option 0)
  const bool dgrad_x3 = x3 && conv1d_x3_supports((const float*)workspace, (int64_t)128 * T, T, 1, (const float*)workspace, T, 128, 64, terms) &&
                        conv1d_x3_supports((const float*)workspace, (int64_t)64 * T, T, 1, (const float*)workspace, T, 64, 32, terms);
  if (!dgrad_x3) {
    DFA_HIP_CHECK(ctx, launch_conv1d_dgrad_pack(p[6], m.wt[0], m.zero_bias, 32, 64, s));
    DFA_HIP_CHECK(ctx, launch_conv1d_dgrad_pack(p[12], m.wt[1], m.zero_bias, 64, 128, s));
  }
  if (x3) DFA_HIP_CHECK(ctx, launch_pack_conv1d_train_all(p[0], p[6], p[12], m.wx3, F, terms, m.zero_bias, s));   // all five images, one launch
  m.train_x3 = x3;
  DropCfg dc{};
  dc.thresh = (p_drop > 0.f) ? (unsigned)((double)p_drop * 4294967296.0) : 0u;
  dc.scale = 1.0f / (1.0f - p_drop);
  dc.seed = seed; dc.offset = offset;
  m.train_drop = dc; m.train_B = B; m.train_T = T;
  char* ws = (char*)workspace;
  float* partial = (float*)(ws + pl.partial);
  const int C[3] = {32, 64, 128}, Cin[3] = {F, 32, 64};
  const int nch = cm_chunks(B);
  for (int l = 0; l < 3; ++l) {
    float* z = (float*)(ws + pl.z[l]);
    const float* const* q = p + 6 * l;
    if (l == 0) {
      if (x3 && conv1d_x3_supports((const float*)x, stride_b, stride_f, stride_t, z, T, F, 32, terms))
        DFA_HIP_CHECK(ctx, launch_conv1d_x3((const float*)x, stride_b, m.wx3[0], q[1], z, B, F, 32, T, terms, s, x3, aug));
      else
        DFA_HIP_CHECK(ctx, launch_conv1d((const float*)x, stride_b, stride_f, stride_t, q[0], q[1], z, B, F, 32, T, false, s, false, aug));
    } else {
      const float* hin = (const float*)(ws + pl.h[l - 1]);
      if (x3 && conv1d_x3_supports(hin, (int64_t)Cin[l] * T, T, 1, z, T, Cin[l], C[l], terms))
        DFA_HIP_CHECK(ctx, launch_conv1d_x3(hin, (int64_t)Cin[l] * T, m.wx3[l], q[1], z, B, Cin[l], C[l], T, terms, s, x3));
      else
        DFA_HIP_CHECK(ctx, launch_conv1d(hin, (int64_t)Cin[l] * T, T, 1, q[0], q[1], z, B, Cin[l], C[l], T, false, s, false));
    }
    St st = st1d(ws, pl, l);
    DFA_HIP_CHECK(ctx, launch_cm_stats(z, partial, B, C[l], T, s));
    { const int rc = finalize_bn_stats(ctx, partial, nch, C[l], (double)B * T, st.mean, st.var, st.invstd,
                                       update_running_stats ? (float*)q[4] : nullptr, update_running_stats ? (float*)q[5] : nullptr, momentum, nullptr);
      if (rc != DFA_OK) return rc; }
    if (l < 2) {
      dc.layer = 1 + l;
      DFA_HIP_CHECK(ctx, launch_cm_bn_relu_drop(z, st.mean, st.invstd, q[2], q[3], (float*)(ws + pl.h[l]), B, C[l], T, dc, s));
    } else {
      DFA_HIP_CHECK(ctx, launch_cm_bn_relu_meant(z, st.mean, st.invstd, q[2], q[3], (float*)(ws + pl.pooled), B, 128, T, s));
    }
  }
  DFA_HIP_CHECK(ctx, launch_linear((const float*)(ws + pl.pooled), p[18], p[19], logits, B, 128, s));
  return DFA_OK;
}

int dfa_cnn1d_backward(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b, int64_t stride_t,
                       int64_t stride_f, const float* dlogits, float* const* grads, int ngrads, void* workspace,
                       size_t workspace_bytes) {
  TraceRange trace_("dfa_cnn1d_backward");
  if (!ctx) return DFA_E_NULL_PTR;
  Cnn1dState& m = ctx->cnn1d;
  if (!m.train_packed || m.train_B != B || m.train_T != T)
    return fail(ctx, DFA_E_NOT_PREPARED, "dfa_cnn1d_backward must follow dfa_cnn1d_forward_train on the same batch");
  if (!x || !dlogits || !grads || !workspace) return fail(ctx, DFA_E_NULL_PTR, "x, dlogits, grads and workspace must be non-null");
  if (x_dtype != DFA_DTYPE_F32) return fail(ctx, DFA_E_BAD_DTYPE, "cnn1d takes float32 input");
  if (ngrads != 14) return fail(ctx, DFA_E_BAD_SHAPE, "cnn1d has 14 parameters, got %d gradient pointers", ngrads);
  for (int i = 0; i < 14; ++i)
    if (!grads[i]) return fail(ctx, DFA_E_NULL_PTR, "gradient pointer %d is null", i);
  const Train1dPlan pl = plan_train1d(B, T, F);
  if (workspace_bytes < pl.total) return fail(ctx, DFA_E_WORKSPACE, "train workspace too small");
  char* ws = (char*)workspace;
  float* partial = (float*)(ws + pl.partial);
  const float* const* p = m.p;
  hipStream_t s = ctx->stream;
  DropCfg dc = m.train_drop;
  const int C[3] = {32, 64, 128}, Cin[3] = {F, 32, 64};
  float* dpooled = (float*)(ws + pl.dpooled);
  DFA_HIP_CHECK(ctx, launch_linear_bwd(dlogits, p[18], (const float*)(ws + pl.pooled), dpooled, grads[12], grads[13], B, 128, s));
  for (int l = 2; l >= 0; --l) {
    const float* const* q = p + 6 * l;
    St st = st1d(ws, pl, l);
    float* sm = sums1d(ws, pl, l);
    float* dz = (float*)(ws + pl.dz[l]);
    const float* up = (l == 2) ? dpooled : (const float*)(ws + pl.dh[l]);
    dc.layer = 1 + l;
    DFA_HIP_CHECK(ctx, launch_cm_bn_bwd(l == 2 ? 0 : 1, (const float*)(ws + pl.z[l]), st.mean, st.invstd, q[2], q[3], up, partial, sm, dz,
                                        B, C[l], T, dc, s, ctx->bn_sync.fn ? &ctx->bn_sync : nullptr));
    hipLaunchKernelGGL(split_sums_kernel, dim3(1), dim3(128), 0, s, sm, grads[4 * l + 2], grads[4 * l + 3], C[l]);
    if (l == 0) {
      DFA_HIP_CHECK(ctx, launch_conv1d_wgrad(dz, (const float*)x, stride_b, stride_f, stride_t, partial, grads[0], grads[1], B, F, 32, T, s,
                                             m.train_aug.on ? &m.train_aug : nullptr, m.train_x3));
    } else {
      DFA_HIP_CHECK(ctx, launch_conv1d_wgrad(dz, (const float*)(ws + pl.h[l - 1]), (int64_t)Cin[l] * T, T, 1, partial, grads[4 * l],
                                             grads[4 * l + 1], B, Cin[l], C[l], T, s, nullptr, m.train_x3));
      // data gradient: dh[l-1] = conv1d(dz; W'[Cin][Cout][3]) -- a Conv1d with Cout input channels, Cin output channels
      float* dh = (float*)(ws + pl.dh[l - 1]);
      const int terms = (m.train_x3 == 3) ? 2 : 3;
      if (m.train_x3 && conv1d_x3_supports(dz, (int64_t)C[l] * T, T, 1, dh, T, C[l], Cin[l], terms))
        DFA_HIP_CHECK(ctx, launch_conv1d_x3(dz, (int64_t)C[l] * T, m.wx3[l == 2 ? 3 : 4], m.zero_bias, dh, B, C[l], Cin[l], T, terms, s, m.train_x3));
      else
        DFA_HIP_CHECK(ctx, launch_conv1d(dz, (int64_t)C[l] * T, T, 1, m.wt[l - 1], m.zero_bias, dh, B, C[l], Cin[l], T, false, s, false));
    }
  }
  DFA_HIP_CHECK(ctx, hipGetLastError());
  return DFA_OK;
}

}  // extern "C"
