// cae_train_api.hip -- C ABI of the ConvAutoencoder training step (replaces, for src/train_cae.py:58-82, torch autograd
// over src/model_cae.py:32-125):  dfa_cae_forward_train (BatchNorm with batch statistics, running-stat update, keeps
// what backward needs in the workspace) and dfa_cae_backward (gradients of the 30 parameters from d(loss)/d(recon)).
#include "dfa_internal.h"
#include "trace.h"
#include "convt2x2_mfma.h"

using namespace dfa;

namespace dfa {
hipError_t launch_cae_train_fwd(int prec, int cin, const ConvArgs& a, float* raw_tmp, hipStream_t s, int wide);
hipError_t launch_cae_dgrad4(int prec, const ConvArgs& a, float* raw_tmp, hipStream_t s, int wide);
hipError_t launch_train_dgrad3(int prec, const ConvArgs& a, float* raw_tmp, hipStream_t s);
hipError_t launch_train_dgrad2(int prec, const ConvArgs& a, hipStream_t s);
__global__ void split_sums_kernel(const float* __restrict__ sums, float* __restrict__ dgamma, float* __restrict__ dbeta, int C);
__global__ void split_c1_kernel(const float* __restrict__ rec, float* __restrict__ dw, float* __restrict__ db);
enum { C1M_STATS = 0, C1M_BWD_REDUCE = 1, C1M_WGRAD = 2 };
enum { SRC_MEANT = 0, SRC_POOL = 1, SRC_DIRECT = 2, SRC_POOL22 = 3 };
}  // namespace dfa

namespace {

inline size_t al(size_t v) { return (v + 255) / 256 * 256; }
constexpr int kWgradWGs = 256, kGemmSplit = 64;
const int kEC[4] = {32, 64, 128, 256};     // encoder block output channels
const int kDC[3] = {128, 64, 32};          // decoder block 1-3 output channels
const int kDCin[3] = {256, 128, 64};

struct CaeTrainPlan {
  int H[5], W[5], Hd[4], Wd[4];
  bool ok;
  size_t e[4], z[4], zd[3], d[3], dd[3], dzd[3], de[4], dz[4], zp, xf, raw, stats, sums, wq, dwq, rec, partial, partial_bytes, total;
};

CaeTrainPlan plan_cae_train(int B, int T, int F, int prec) {
  CaeTrainPlan p;
  const size_t es = (prec == DFA_PREC_BF16) ? 2 : 4;
  p.H[0] = T; p.W[0] = F;
  for (int l = 1; l <= 4; ++l) { p.H[l] = p.H[l - 1] / 2; p.W[l] = p.W[l - 1] / 2; }
  p.Hd[0] = 2 * p.H[4]; p.Wd[0] = 2 * p.W[4];
  p.Hd[1] = 2 * p.Hd[0]; p.Wd[1] = 2 * p.Wd[0] + 1;
  p.Hd[2] = 2 * p.Hd[1]; p.Wd[2] = 2 * p.Wd[1];
  p.Hd[3] = 2 * p.Hd[2]; p.Wd[3] = 2 * p.Wd[2];
  p.ok = (p.H[4] >= 1 && p.W[4] >= 1 && p.Wd[3] == F);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = al(off + bytes); return o; };
  for (int l = 0; l < 4; ++l) p.e[l] = take((size_t)B * p.H[l + 1] * p.W[l + 1] * kEC[l] * es);
  p.z[0] = 0;  // block 1's pre-BN output is recomputed, never stored
  for (int l = 1; l < 4; ++l) p.z[l] = take((size_t)B * p.H[l] * p.W[l] * kEC[l] * es);
  for (int l = 0; l < 3; ++l) p.zd[l] = take((size_t)B * p.Hd[l] * p.Wd[l] * kDC[l] * es);
  for (int l = 0; l < 3; ++l) p.d[l] = take((size_t)B * p.Hd[l] * p.Wd[l] * kDC[l] * es);
  for (int l = 0; l < 3; ++l) p.dd[l] = take((size_t)B * p.Hd[l] * p.Wd[l] * kDC[l] * es);
  for (int l = 0; l < 3; ++l) p.dzd[l] = take((size_t)B * p.Hd[l] * p.Wd[l] * kDC[l] * es);
  for (int l = 0; l < 4; ++l) p.de[l] = take((size_t)B * p.H[l + 1] * p.W[l + 1] * kEC[l] * es);
  p.dz[0] = 0;
  for (int l = 1; l < 4; ++l) p.dz[l] = take((size_t)B * p.H[l] * p.W[l] * kEC[l] * es);
  size_t zp = 0, xf = 0;
  for (int l = 0; l < 3; ++l) {
    const size_t P = (size_t)B * (p.Hd[l] / 2) * (l == 1 ? (p.Wd[l] - 1) / 2 : p.Wd[l] / 2);
    zp = std::max(zp, P * 4 * kDC[l] * es);
    xf = std::max(xf, P * kDCin[l] * 4);
  }
  p.zp = take(zp);
  p.xf = take(xf);
  size_t raw = (size_t)B * p.H[3] * p.W[3] * 256 * 4;                    // enc4 forward split
  raw = std::max(raw, (size_t)B * p.H[2] * p.W[2] * 64 * 4);             // dgrad3
  p.raw = take(raw);
  p.stats = take(704 * 3 * 4);
  p.sums = take(704 * 2 * 4 + 1024 * 4);
  p.wq = take((size_t)256 * 512 * 4);
  p.dwq = take((size_t)256 * 512 * 4);
  p.rec = take(1024 * 4);
  size_t pb = (size_t)kWgradWGs * ((size_t)128 * 64 * 9 + 256) * 4;
  pb = std::max(pb, (size_t)kGemmSplit * 256 * 512 * 4);
  pb = std::max(pb, ((size_t)conv1_train_blocks(B, T, F) + 64) * 320 * 4);
  int ppb;
  pb = std::max(pb, (size_t)cl_stats_blocks((size_t)B * T * F, &ppb) * 256 * 2 * 4);
  pb = std::max(pb, (size_t)cae_dec4_bwd_blocks() * 132 * 4);
  // channel-sum records of the decoder's folded BatchNorm-backward apply pass (cae_bwd_fold): one [C] record per 16 * (256 / (C / 8)) pixels
  for (int l = 0; l < 3; ++l) {
    const size_t npix = (size_t)B * p.Hd[l] * p.Wd[l], ppb2 = 16 * (256 / (kDC[l] / 8));
    pb = std::max(pb, 2 * ((npix + ppb2 - 1) / ppb2 + 64) * kDC[l] * 4);
  }
  // statistics records of the convolution epilogues (cae_conv_stats): the lower half holds the records, the upper half the second
  // reduction level of the synchronised form
  for (int l = 1; l < 4; ++l) pb = std::max(pb, 2 * ((size_t)B * ((p.W[l] + 31) / 32) + 64) * kEC[l] * 2 * 4);
  for (int l = 0; l < 3; ++l) {
    const long P = (long)B * (p.Hd[l] / 2) * (l == 1 ? (p.Wd[l] - 1) / 2 : p.Wd[l] / 2);
    pb = std::max(pb, 2 * ((size_t)cae_dec_stats_records(prec, kDCin[l], P) + 64) * kDC[l] * 2 * 4);
  }
  p.partial = take(pb);
  p.partial_bytes = pb;
  p.total = off;
  return p;
}

// BN layer order in the stats / sums blocks: enc1..enc4 (32, 64, 128, 256), dec1..dec3 (128, 64, 32)
const int kBnOff[7] = {0, 32, 96, 224, 480, 608, 672};
struct St { float *mean, *var, *invstd; };
St stat_of(char* ws, const CaeTrainPlan& pl, int layer, int C) {
  float* b = (float*)(ws + pl.stats) + 3 * kBnOff[layer];
  return {b, b + C, b + 2 * C};
}
float* sums_of(char* ws, const CaeTrainPlan& pl, int layer) { return (float*)(ws + pl.sums) + 2 * kBnOff[layer]; }

// batch statistics from the block records partial[nparts][C][2]; under synchronised BatchNorm (dfa_ctx_set_bn_sync) the records are
// reduced to one [C][2] record in the caller's buffer, summed over the ranks by the hook, and the global count is used
int finalize_records(dfa_ctx* ctx, const float* partial, int nparts, int C, double n, const St& st, float* rm, float* rv, float momentum,
                     float* scratch = nullptr) {
  const dfa::BnSync& sy = ctx->bn_sync;
  if (!sy.fn) {
    DFA_HIP_CHECK(ctx, launch_bn_finalize(partial, nparts, C, n, st.mean, st.var, st.invstd, rm, rv, momentum, ctx->stream));
    return DFA_OK;
  }
  DFA_HIP_CHECK(ctx, launch_reduce_partials(partial, nparts, C * 2, 1.0f, sy.buf, ctx->stream, scratch));
  if (sy.fn(sy.user, sy.buf, C * 2) != 0) return fail(ctx, DFA_E_HIP, "the BatchNorm synchronisation hook failed (forward statistics, %d channels)", C);
  DFA_HIP_CHECK(ctx, launch_bn_finalize(sy.buf, 1, C, n * (double)sy.world, st.mean, st.var, st.invstd, rm, rv, momentum, ctx->stream));
  return DFA_OK;
}

int finalize_stats(dfa_ctx* ctx, int prec, const void* z, size_t npix, int C, const St& st, float* partial, float* rm,
                   float* rv, float momentum) {
  int ppb;
  const int nblk = cl_stats_blocks(npix, &ppb);
  DFA_HIP_CHECK(ctx, launch_cl_stats(prec, z, partial, npix, C, ctx->stream));
  return finalize_records(ctx, partial, nblk, C, (double)npix, st, rm, rv, momentum);
}

}  // namespace

extern "C" {

size_t dfa_cae_train_workspace_bytes(const dfa_ctx* ctx, int B, int T, int F, int precision) {
  (void)ctx;
  if (B < 1 || T < 16 || F < 16) return 0;
  const CaeTrainPlan pl = plan_cae_train(B, T, F, precision);
  return pl.ok ? pl.total : 0;
}

int dfa_cae_forward_train(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b, int64_t stride_t,
                          int64_t stride_f, int precision, float momentum, int update_running_stats, float* recon,
                          float* latent, float* mse, void* workspace, size_t workspace_bytes) {
  TraceRange trace_("dfa_cae_forward_train");
  if (!ctx) return DFA_E_NULL_PTR;
  CaeState& m = ctx->cae;
  if (!m.have_params) return fail(ctx, DFA_E_NOT_PREPARED, "dfa_cae_set_params has not been called");
  if (!x || !workspace) return fail(ctx, DFA_E_NULL_PTR, "x and workspace must be non-null");
  if (!recon && !mse) return fail(ctx, DFA_E_NULL_PTR, "give recon, mse or both (the decoder's last kernel needs an output)");
  if (x_dtype != DFA_DTYPE_F32 && x_dtype != DFA_DTYPE_BF16) return fail(ctx, DFA_E_BAD_DTYPE, "x dtype %d not supported", x_dtype);
  if (precision != DFA_PREC_F32 && precision != DFA_PREC_BF16) return fail(ctx, DFA_E_BAD_DTYPE, "unknown precision %d", precision);
  if (B < 1 || T < 16) return fail(ctx, DFA_E_BAD_SHAPE, "need B >= 1 and T >= 16 (got %d, %d)", B, T);
  const CaeTrainPlan pl = plan_cae_train(B, T, F, precision);
  if (!pl.ok) return fail(ctx, DFA_E_BAD_SHAPE, "F=%d: decoder would rebuild %d columns (needs F = 16*(F/16)+4)", F, pl.Wd[3]);
  if (workspace_bytes < pl.total) return fail(ctx, DFA_E_WORKSPACE, "train workspace too small: %zu < %zu bytes", workspace_bytes, pl.total);
  DFA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int prec = precision;
  if (!m.train_packed) {  // raw conv images enc2-4 (+ their dgrad images), raw convT images, conv1 fold target, biases
    size_t off = al((288 + 32) * 4);
    size_t eb[3], db[3], ew[3], dgw[3], dw[3], dgb;
    for (int l = 0; l < 3; ++l) { eb[l] = off; off = al(off + kEC[l + 1] * 4); }
    for (int l = 0; l < 3; ++l) { db[l] = off; off = al(off + kDC[l] * 4); }
    dgb = off; off = al(off + 256 * 4);
    for (int l = 0; l < 3; ++l) { ew[l] = off; off = al(off + (size_t)kEC[l + 1] * kEC[l] * 9 * 4); }
    for (int l = 0; l < 3; ++l) { dgw[l] = off; off = al(off + (size_t)kEC[l + 1] * kEC[l] * 9 * 4); }
    for (int l = 0; l < 3; ++l) { dw[l] = off; off = al(off + (size_t)kDC[l] * kDCin[l] * 4 * 4); }
    DFA_HIP_CHECK(ctx, hipMalloc(&m.train_packed, off));
    char* b = (char*)m.train_packed;
    m.tw1 = (float*)b; m.tb1 = m.tw1 + 288;
    for (int l = 0; l < 3; ++l) {
      m.tenc[l].bias = (float*)(b + eb[l]); m.tenc[l].wpack = (uint4*)(b + ew[l]);
      m.tdg[l].bias = (float*)(b + dgb); m.tdg[l].wpack = (uint4*)(b + dgw[l]);
      m.tdec[l].bias = (float*)(b + db[l]); m.tdec[l].wpack = (uint4*)(b + dw[l]);
    }
  }
  const float* const* p = m.p;
  hipStream_t s = ctx->stream;
  char* ws = (char*)workspace;
  float* partial = (float*)(ws + pl.partial);
  const int nkg = (prec == DFA_PREC_BF16) ? 4 : 8;
  // ---- weight images (weights move every step)
  DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3(p[6], p[7], nullptr, nullptr, nullptr, nullptr, 32, 0, 32, 64, prec, m.tenc[0].wpack, m.tenc[0].bias, s, 0));
  DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3(p[12], p[13], nullptr, nullptr, nullptr, nullptr, 64, 0, 64, 128, prec, m.tenc[1].wpack, m.tenc[1].bias, s, 0));
  m.train_enc4_wide = (prec == DFA_PREC_BF16 && ctx->cae_enc4_wide) ? 1 : 0;
  if (m.train_enc4_wide) {   // one 128-input-channel image (the eval forward's layout)
    DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3(p[18], p[19], nullptr, nullptr, nullptr, nullptr, 128, 0, 128, 256, prec, m.tenc[2].wpack, m.tenc[2].bias, s, 0));
  } else {
  for (int hlf = 0; hlf < 2; ++hlf)
    DFA_HIP_CHECK(ctx, launch_fold_pack_conv3x3(p[18], p[19], nullptr, nullptr, nullptr, nullptr, 128, 64 * hlf, 64, 256, prec,
                                                m.tenc[2].wpack + (size_t)hlf * (256 / 32) * 9 * nkg * 64, m.tenc[2].bias, s, 0));
  }
  // bf16 mode: the 64 -> 32 and 128 -> 64 data gradients on the 16x16x32 kernels of conv_split.hip, one launch each (as the CNN2D's;
  // the 32x32x16 forms ran the first with one channel slice per workgroup -- 0.31 ms -- and the second as two launches chained
  // through fp32 partial sums)
  m.train_dgrad_m16 = (prec == DFA_PREC_BF16 && ctx->dgrad_m16) ? 1 : 0;
  if (m.train_dgrad_m16) {
    DFA_HIP_CHECK(ctx, launch_pack_conv3x3_dgrad_m16(p[6], 32, 64, m.tdg[0].wpack, m.tdg[0].bias, s));
    DFA_HIP_CHECK(ctx, launch_pack_conv3x3_dgrad_m16(p[12], 64, 128, m.tdg[1].wpack, m.tdg[1].bias, s));
  } else {
  DFA_HIP_CHECK(ctx, launch_pack_conv3x3_dgrad(p[6], 32, 64, 0, 64, prec, m.tdg[0].wpack, m.tdg[0].bias, s));
  for (int hlf = 0; hlf < 2; ++hlf)
    DFA_HIP_CHECK(ctx, launch_pack_conv3x3_dgrad(p[12], 64, 128, 64 * hlf, 64, prec, m.tdg[1].wpack + (size_t)hlf * (64 / 32) * 9 * nkg * 64, m.tdg[1].bias, s));
  }
  if (m.train_enc4_wide) {
    for (int c = 0; c < 2; ++c)
      DFA_HIP_CHECK(ctx, launch_pack_conv3x3_dgrad(p[18], 128, 256, 128 * c, 128, prec, m.tdg[2].wpack + (size_t)c * (128 / 32) * 9 * 8 * 64, m.tdg[2].bias, s));
  } else {
  for (int c = 0; c < 4; ++c)
    DFA_HIP_CHECK(ctx, launch_pack_conv3x3_dgrad(p[18], 128, 256, 64 * c, 64, prec, m.tdg[2].wpack + (size_t)c * (128 / 32) * 9 * nkg * 64, m.tdg[2].bias, s));
  }
  for (int l = 0; l < 3; ++l) {
    const float* const* q = p + 24 + 6 * l;
    DFA_HIP_CHECK(ctx, launch_fold_pack_convt2x2(q[0], q[1], nullptr, nullptr, nullptr, nullptr, kDCin[l], kDC[l], prec, m.tdec[l].wpack, m.tdec[l].bias, s, 0));
  }
  m.train_prec = prec; m.train_B = B; m.train_T = T;
  auto rmv = [&](int pi) { return update_running_stats ? (float*)p[pi] : nullptr; };
  DropCfg nodrop{};
  // ---- encoder block 1 (pre-BN output recomputed from x: statistics pass, fold, fused eval kernel)
  {
    St st = stat_of(ws, pl, 0, 32);
    // bf16 mode on bf16 features (F even): the statistics pass (with the 9 x 9 tap moments the fused backward algebra needs) and
    // the backward pass on the matrix cores (train_conv1_mfma.hip, as the CNN2D's block 1), the forward on cae_enc1_mfma.hip
    // (synchronised BatchNorm: the backward needs the layer's sums before the weight gradient is formed -> the two-pass vector path)
    m.train_c1_mfma = (ctx->conv1_mfma && !ctx->bn_sync.fn && prec == DFA_PREC_BF16 && x_dtype == DFA_DTYPE_BF16 && F <= 224 && !(F & 1) && T >= 4) ? 1 : 0;
    if (m.train_c1_mfma) {
      const int nbm = conv1_mfma_blocks(B, T, F);
      DFA_HIP_CHECK(ctx, launch_conv1_mfma(C1X_STATS, x, stride_b, stride_t, stride_f, p[0], p[1], nullptr, nullptr, partial, B, T, F, nodrop, s));
      DFA_HIP_CHECK(ctx, launch_bn_finalize(partial, nbm, 32, (double)B * T * F, st.mean, st.var, st.invstd, rmv(4), rmv(5), momentum, s));
      float* xxs = (float*)(ws + pl.rec) + 512;           // XX[9][9] | Xs[9]: kept for the backward
      DFA_HIP_CHECK(ctx, launch_reduce_partials(partial + (size_t)nbm * 64, nbm, 96, 1.0f, xxs, s, partial + (size_t)nbm * 160));
    } else {
      DFA_HIP_CHECK(ctx, launch_conv1_train(C1M_STATS, x, x_dtype, stride_b, stride_t, stride_f, p[0], p[1], nullptr, nullptr, nullptr, nullptr,
                                            nullptr, nullptr, prec, partial, B, T, F, nodrop, s));
      { const int rc = finalize_records(ctx, partial, conv1_train_blocks(B, T, F), 32, (double)B * T * F, st, rmv(4), rmv(5), momentum);
        if (rc != DFA_OK) return rc; }
    }
    DFA_HIP_CHECK(ctx, launch_fold_conv1(p[0], p[1], p[2], p[3], st.mean, st.var, m.tw1, m.tb1, 32, s));
    if (m.train_c1_mfma && ctx->cae_enc1_mfma) {
      uint4* tpack = (uint4*)(ws + pl.wq);                // this step's three-term A operands (pl.wq is free until the decoder's backward)
      float* tbias = (float*)(ws + pl.wq) + 6 * 64 * 4;
      DFA_HIP_CHECK(ctx, launch_pack_cae_enc1_mfma(m.tw1, m.tb1, tpack, tbias, s));
      DFA_HIP_CHECK(ctx, launch_cae_enc1_mfma(x, x_dtype, stride_b, stride_t, stride_f, nullptr, nullptr, tpack, tbias, ws + pl.e[0], B, T, F, s));
    } else {
      DFA_HIP_CHECK(ctx, launch_cae_enc1(x, x_dtype, stride_b, stride_t, stride_f, nullptr, nullptr, m.tw1, m.tb1, ws + pl.e[0], prec, B, T, F, s));
    }
  }
  // ---- encoder blocks 2-4
  for (int l = 1; l < 4; ++l) {
    ConvArgs a{};
    a.in = ws + pl.e[l - 1]; a.wpack = m.tenc[l - 1].wpack; a.bias = m.tenc[l - 1].bias; a.out = ws + pl.z[l];
    a.B = B; a.H = pl.H[l]; a.W = pl.W[l]; a.COUT = kEC[l]; a.relu = 0; a.zero_page = ctx->zero_page;
    const bool epi_stats = ctx->cae_conv_stats && l < 3;         // (block 4: conv3x3_inst_cae_train.hip)
    a.stats_partial = epi_stats ? partial : nullptr;             // one [COUT][2] record per (sample, 32-column strip)
    DFA_HIP_CHECK(ctx, launch_cae_train_fwd(prec, kEC[l - 1], a, (float*)(ws + pl.raw), s, m.train_enc4_wide));
    St st = stat_of(ws, pl, l, kEC[l]);
    const size_t npix = (size_t)B * pl.H[l] * pl.W[l];
    int rc = epi_stats ? finalize_records(ctx, partial, B * ((pl.W[l] + 31) / 32), kEC[l], (double)npix, st, rmv(6 * l + 4), rmv(6 * l + 5), momentum)
                                 : finalize_stats(ctx, prec, ws + pl.z[l], npix, kEC[l], st, partial, rmv(6 * l + 4), rmv(6 * l + 5), momentum);
    if (rc != DFA_OK) return rc;
    DFA_HIP_CHECK(ctx, launch_bn_relu_pool(prec, 2, ws + pl.z[l], st.mean, st.invstd, p[6 * l + 2], p[6 * l + 3], ws + pl.e[l], B, pl.H[l], pl.W[l], kEC[l], s));
  }
  if (latent) DFA_HIP_CHECK(ctx, launch_cae_latent_export(ws + pl.e[3], prec, latent, B, pl.H[4] * pl.W[4], 256, s));
  // ---- decoder blocks 1-3: raw ConvTranspose2d -> statistics -> BN + ReLU
  for (int l = 0; l < 3; ++l) {
    const float* const* q = p + 24 + 6 * l;
    ConvTArgs a{};
    a.in = (l == 0) ? ws + pl.e[3] : ws + pl.d[l - 1];
    a.wpack = m.tdec[l].wpack; a.bias = m.tdec[l].bias; a.out = ws + pl.zd[l];
    a.B = B; a.H = pl.Hd[l] / 2; a.W = (l == 1) ? (pl.Wd[l] - 1) / 2 : pl.Wd[l] / 2;
    a.COUT = kDC[l]; a.opad_w = (l == 1) ? 1 : 0; a.no_relu = 1;
    a.stats_partial = ctx->cae_conv_stats ? partial : nullptr;   // one [COUT][2] record per workgroup (the padding column's share in record 0)
    DFA_HIP_CHECK(ctx, launch_cae_dec(prec, kDCin[l], a, s));
    if (l == 1) DFA_HIP_CHECK(ctx, launch_cae_opad_col(ws + pl.zd[1], m.tdec[1].bias, prec, B * pl.Hd[1], pl.Wd[1], 64, s, 1));
    St st = stat_of(ws, pl, 4 + l, kDC[l]);
    const size_t npix = (size_t)B * pl.Hd[l] * pl.Wd[l];
    int rc = ctx->cae_conv_stats ? finalize_records(ctx, partial, cae_dec_stats_records(prec, kDCin[l], (long)B * a.H * a.W), kDC[l], (double)npix, st,
                                                    rmv(24 + 6 * l + 4), rmv(24 + 6 * l + 5), momentum, partial + pl.partial_bytes / 8)
                                 : finalize_stats(ctx, prec, ws + pl.zd[l], npix, kDC[l], st, partial, rmv(24 + 6 * l + 4), rmv(24 + 6 * l + 5), momentum);
    if (rc != DFA_OK) return rc;
    DFA_HIP_CHECK(ctx, launch_bn_relu_pool(prec, 1, ws + pl.zd[l], st.mean, st.invstd, q[2], q[3], ws + pl.d[l], B, pl.Hd[l], pl.Wd[l], kDC[l], s));
  }
  // ---- decoder block 4 + zero time padding (+ per-sample MSE)
  DFA_HIP_CHECK(ctx, launch_cae_dec4_mse(ws + pl.d[2], prec, p[42], p[43], x, x_dtype, stride_b, stride_t, stride_f, nullptr, nullptr, recon,
                                         partial, mse, B, pl.Hd[2], pl.Wd[2], T, F, s));
  return DFA_OK;
}

int dfa_cae_backward(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b, int64_t stride_t,
                     int64_t stride_f, const float* drecon, float* const* grads, int ngrads, void* workspace,
                     size_t workspace_bytes) {
  TraceRange trace_("dfa_cae_backward");
  if (!ctx) return DFA_E_NULL_PTR;
  CaeState& m = ctx->cae;
  if (!m.train_packed || m.train_B != B || m.train_T != T)
    return fail(ctx, DFA_E_NOT_PREPARED, "dfa_cae_backward must follow dfa_cae_forward_train on the same batch");
  if (!x || !grads || !workspace) return fail(ctx, DFA_E_NULL_PTR, "x, grads and workspace must be non-null");
  if (ngrads != 30) return fail(ctx, DFA_E_BAD_SHAPE, "the auto-encoder has 30 parameters, got %d gradient pointers", ngrads);
  for (int i = 0; i < 30; ++i)
    if (!grads[i]) return fail(ctx, DFA_E_NULL_PTR, "gradient pointer %d is null", i);
  const int prec = m.train_prec;
  const CaeTrainPlan pl = plan_cae_train(B, T, F, prec);
  if (workspace_bytes < pl.total) return fail(ctx, DFA_E_WORKSPACE, "train workspace too small");
  const float* const* p = m.p;
  hipStream_t s = ctx->stream;
  char* ws = (char*)workspace;
  float* partial = (float*)(ws + pl.partial);
  float* rec = (float*)(ws + pl.rec);
  float* scratch_c = (float*)(ws + pl.sums) + 2 * 704;   // 1024 floats of scratch behind the sums block
  // second-level scratch of the block-record reductions (64 x C x 2 floats): the upper half of the partial buffer -- the records
  // of the BatchNorm / statistics passes (<= 3600 x 512 floats) use a fraction of the lower half (sized for the weight gradients)
  float* scratch2 = partial + pl.partial_bytes / 8;
  const dfa::BnSync* sync = ctx->bn_sync.fn ? &ctx->bn_sync : nullptr;     // synchronised BatchNorm (dfa_ctx_set_bn_sync)
  const int bf = (prec == DFA_PREC_BF16) ? 1 : 0;
  DropCfg nodrop{};
  // ---- decoder block 4
  // drecon == NULL: the loss is MSELoss(recon, x) (src/train_cae.py:67-68) and its gradient 2 (recon - x) / (B T F) is formed inside
  // the kernel from the saved d3 and x -- neither recon nor drecon has to exist in memory
  MseArgs ma{x, x_dtype == DFA_DTYPE_BF16 ? 1 : 0, stride_b, stride_t, stride_f, p[43]};
  DFA_HIP_CHECK(ctx, launch_cae_dec4_bwd(prec, ws + pl.d[2], p[42], drecon, ws + pl.dd[2], partial, B, pl.Hd[2], pl.Wd[2], T, F, s, drecon ? nullptr : &ma));
  DFA_HIP_CHECK(ctx, launch_reduce_partials(partial, cae_dec4_bwd_blocks(), 132, 1.0f, rec, s, nullptr));
  DFA_HIP_CHECK(ctx, hipMemcpyAsync(grads[28], rec, 128 * 4, hipMemcpyDeviceToDevice, s));
  DFA_HIP_CHECK(ctx, hipMemcpyAsync(grads[29], rec + 128, 4, hipMemcpyDeviceToDevice, s));
  // ---- decoder blocks 3, 2, 1
  for (int l = 2; l >= 0; --l) {
    const float* const* q = p + 24 + 6 * l;
    const int Hin = pl.Hd[l] / 2, Win = (l == 1) ? (pl.Wd[l] - 1) / 2 : pl.Wd[l] / 2;
    const int Cin = kDCin[l], Cout = kDC[l];
    const long P = (long)B * Hin * Win;
    St st = stat_of(ws, pl, 4 + l, Cout);
    float* sm = sums_of(ws, pl, 4 + l);
    if (ctx->cae_bwd_fold) {
      // one apply pass writes dz patch-major (what the two gradient GEMMs read) and leaves the records of its channel sums = the
      // ConvTranspose2d bias gradient (over ALL output pixels, the output_padding column included): dzd itself is never stored
      dfa::BnBwdFold fold{ws + pl.zp, Win, partial, 0};
      DFA_HIP_CHECK(ctx, launch_bn_bwd(prec, SRC_DIRECT, ws + pl.zd[l], st.mean, st.invstd, q[2], q[3], nullptr, ws + pl.dd[l], partial, sm,
                                       nullptr, B, pl.Hd[l], pl.Wd[l], Cout, nodrop, s, scratch2, sync, &fold));
      hipLaunchKernelGGL(split_sums_kernel, dim3(1), dim3(256), 0, s, sm, grads[16 + 4 * l + 2], grads[16 + 4 * l + 3], Cout);
      DFA_HIP_CHECK(ctx, launch_reduce_partials(partial, fold.nrec, Cout, 1.0f, grads[16 + 4 * l + 1], s, scratch2));
    } else {
    DFA_HIP_CHECK(ctx, launch_bn_bwd(prec, SRC_DIRECT, ws + pl.zd[l], st.mean, st.invstd, q[2], q[3], nullptr, ws + pl.dd[l], partial, sm,
                                     ws + pl.dzd[l], B, pl.Hd[l], pl.Wd[l], Cout, nodrop, s, scratch2, sync));
    hipLaunchKernelGGL(split_sums_kernel, dim3(1), dim3(256), 0, s, sm, grads[16 + 4 * l + 2], grads[16 + 4 * l + 3], Cout);
    {  // ConvTranspose2d bias gradient = channel sums of dz over ALL output pixels (the output_padding column included)
      int ppb;
      const size_t npix = (size_t)B * pl.Hd[l] * pl.Wd[l];
      const int nblk = cl_stats_blocks(npix, &ppb);
      DFA_HIP_CHECK(ctx, launch_cl_stats(prec, ws + pl.dzd[l], partial, npix, Cout, s));
      DFA_HIP_CHECK(ctx, launch_reduce_partials(partial, nblk, Cout * 2, 1.0f, scratch_c, s, scratch2));
      hipLaunchKernelGGL(split_sums_kernel, dim3(1), dim3(256), 0, s, scratch_c, scratch_c + 512, grads[16 + 4 * l + 1], Cout);
    }
    DFA_HIP_CHECK(ctx, launch_pixel_unshuffle(prec, ws + pl.dzd[l], ws + pl.zp, B, Hin, Win, pl.Wd[l], Cout, s));
    }
    float* wq = (float*)(ws + pl.wq);
    DFA_HIP_CHECK(ctx, launch_convt_w_to_q(q[0], wq, Cin, Cout, s));
    // data gradient: dX[P x Cin] = Zp[P x 4Cout] . Wq^T
    float* xf = (float*)(ws + pl.xf);
    void* dx = (l == 0) ? ws + pl.de[3] : ws + pl.dd[l - 1];
    if (bf && ctx->cae_dgrad_mfma && convt_dgrad_bf16_supports(Cin, Cout)) {   // bf16 matrix cores, bf16 result in place
      DFA_HIP_CHECK(ctx, launch_convt_dgrad_bf16(ws + pl.zp, wq, xf, dx, P, Cin, Cout, s));   // (xf: scratch for the bf16 weight fragments)
    } else {
      DFA_HIP_CHECK(ctx, launch_gemm_f32(bf, ws + pl.zp, 4 * Cout, 1, 0, wq, 1, 4 * Cout, xf, (int)P, Cin, 4 * Cout, 1, s));
      DFA_HIP_CHECK(ctx, launch_cast_from_f32(prec, xf, dx, (size_t)P * Cin, s));
    }
    // weight gradient: dWq[Cin x 4Cout] = X^T . Zp   (K = P, split over workgroups)
    const void* xin = (l == 0) ? ws + pl.e[3] : ws + pl.d[l - 1];
    float* dwq = (float*)(ws + pl.dwq);
    int nparts = kGemmSplit;
    if (bf) {   // bf16 mode: transposed-read bf16 MFMA GEMM (gemm_tn_bf16.hip)
      DFA_HIP_CHECK(ctx, launch_gemm_tn_bf16(xin, ws + pl.zp, partial, pl.partial_bytes / sizeof(float), Cin, 4 * Cout, (int)P, &nparts, s));
    } else {
      DFA_HIP_CHECK(ctx, launch_gemm_f32(bf, xin, 1, Cin, bf, ws + pl.zp, 4 * Cout, 1, partial, Cin, 4 * Cout, (int)P, kGemmSplit, s));
    }
    DFA_HIP_CHECK(ctx, launch_reduce_partials(partial, nparts, Cin * 4 * Cout, 1.0f, dwq, s, nullptr));
    DFA_HIP_CHECK(ctx, launch_convt_q_to_w(dwq, grads[16 + 4 * l], Cin, Cout, s));
  }
  // ---- encoder blocks 4, 3, 2
  for (int l = 3; l >= 1; --l) {
    St st = stat_of(ws, pl, l, kEC[l]);
    float* sm = sums_of(ws, pl, l);
    DFA_HIP_CHECK(ctx, launch_bn_bwd(prec, SRC_POOL22, ws + pl.z[l], st.mean, st.invstd, p[6 * l + 2], p[6 * l + 3], nullptr, ws + pl.de[l], partial, sm,
                                     ws + pl.dz[l], B, pl.H[l], pl.W[l], kEC[l], nodrop, s, scratch2, sync));
    hipLaunchKernelGGL(split_sums_kernel, dim3(1), dim3(256), 0, s, sm, grads[4 * l + 2], grads[4 * l + 3], kEC[l]);
    if (l == 3) {
      for (int co = 0; co < 2; ++co)
        for (int ci = 0; ci < 2; ++ci)
          DFA_HIP_CHECK(ctx, launch_wgrad3x3_window(prec, 64, 128, 128, 256, 64 * ci, 128 * co, ws + pl.dz[3], ws + pl.e[2], partial, grads[12],
                                                    ci == 0 ? grads[13] : nullptr, B, pl.H[3], pl.W[3], kWgradWGs, s));
    } else {
      DFA_HIP_CHECK(ctx, launch_wgrad3x3(prec, kEC[l - 1], kEC[l], ws + pl.dz[l], ws + pl.e[l - 1], partial, grads[4 * l], grads[4 * l + 1], B,
                                         pl.H[l], pl.W[l], kWgradWGs, s));
    }
    ConvArgs a{};
    a.in = ws + pl.dz[l]; a.wpack = m.tdg[l - 1].wpack; a.bias = m.tdg[l - 1].bias; a.out = ws + pl.de[l - 1];
    a.B = B; a.H = pl.H[l]; a.W = pl.W[l]; a.COUT = kEC[l - 1]; a.relu = 0; a.zero_page = ctx->zero_page;
    hipError_t e = (l == 3) ? launch_cae_dgrad4(prec, a, (float*)(ws + pl.raw), s, m.train_enc4_wide)
                 : (l == 2) ? (m.train_dgrad_m16 ? launch_train_dgrad3_m16(a, s, train_conv_variant() != 0) : launch_train_dgrad3(prec, a, (float*)(ws + pl.raw), s))
                            : (m.train_dgrad_m16 ? launch_train_dgrad2_m16(a, s, train_conv_variant() != 0) : launch_train_dgrad2(prec, a, s));
    DFA_HIP_CHECK(ctx, e);
  }
  // ---- encoder block 1 (z1 recomputed from x; upstream through the 2x2 average pool)
  {
    St st = stat_of(ws, pl, 0, 32);
    float* sm = sums_of(ws, pl, 0);
    const int nb1 = conv1_train_blocks(B, T, F);
    float* scratch = partial + (size_t)nb1 * 320;
    if (m.train_c1_mfma && ctx->conv1_mfma) {   // one pass on the matrix cores + the fused algebra (train_conv1.hip header)
      const int nbm = conv1_mfma_blocks(B, T, F);
      float* xxs = (float*)(ws + pl.rec) + 512;
      float* c1rec = (float*)(ws + pl.rec) + 640;         // [32][11]
      DFA_HIP_CHECK(ctx, launch_conv1_mfma(C1X_BWD, x, stride_b, stride_t, stride_f, m.tw1, m.tb1, nullptr, ws + pl.de[0], partial, B, T, F, nodrop, s, 2));
      DFA_HIP_CHECK(ctx, launch_reduce_partials(partial, nbm, 352, 1.0f, c1rec, s, partial + (size_t)nbm * 352));
      DFA_HIP_CHECK(ctx, launch_conv1_bwd_finalize(c1rec, xxs, p[0], p[1], st.mean, st.invstd, p[2], (double)B * T * F, grads[0], grads[1],
                                                   grads[2], grads[3], s, 1));
      DFA_HIP_CHECK(ctx, hipGetLastError());
      return DFA_OK;
    }
    DFA_HIP_CHECK(ctx, launch_conv1_train(C1M_BWD_REDUCE, x, x_dtype, stride_b, stride_t, stride_f, p[0], p[1], st.mean, st.invstd, p[2], p[3],
                                          nullptr, ws + pl.de[0], prec, partial, B, T, F, nodrop, s, 2));
    DFA_HIP_CHECK(ctx, launch_reduce_partials(partial, nb1, 64, 1.0f, sm, s, scratch));
    hipLaunchKernelGGL(split_sums_kernel, dim3(1), dim3(256), 0, s, sm, grads[2], grads[3], 32);      // dgamma, dbeta: this rank's own sums
    const float* sm_a;
    float isc;
    DFA_HIP_CHECK(ctx, bn_sync_sums(sync, sm, 64, s, &sm_a, &isc));                                    // dz1 is formed from the global ones
    DFA_HIP_CHECK(ctx, launch_conv1_train(C1M_WGRAD, x, x_dtype, stride_b, stride_t, stride_f, p[0], p[1], st.mean, st.invstd, p[2], p[3],
                                          sm_a, ws + pl.de[0], prec, partial, B, T, F, nodrop, s, 2, nullptr, isc));
    DFA_HIP_CHECK(ctx, launch_reduce_partials(partial, nb1, 320, 1.0f, rec, s, scratch));
    hipLaunchKernelGGL(split_c1_kernel, dim3(1), dim3(320), 0, s, rec, grads[0], grads[1]);
  }
  DFA_HIP_CHECK(ctx, hipGetLastError());
  return DFA_OK;
}

int dfa_mse_fwd_bwd(dfa_ctx* ctx, const float* recon, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b,
                    int64_t stride_t, int64_t stride_f, float* loss, float* drecon) {
  TraceRange trace_("dfa_mse_fwd_bwd");
  if (!ctx) return DFA_E_NULL_PTR;
  if (!recon || !x) return fail(ctx, DFA_E_NULL_PTR, "recon and x must be non-null");
  if (!loss && !drecon) return fail(ctx, DFA_E_NULL_PTR, "give loss, drecon or both");
  if (x_dtype != DFA_DTYPE_F32 && x_dtype != DFA_DTYPE_BF16) return fail(ctx, DFA_E_BAD_DTYPE, "x dtype %d not supported", x_dtype);
  if (B < 1 || T < 1 || F < 1) return fail(ctx, DFA_E_BAD_SHAPE, "need B, T, F >= 1 (got %d, %d, %d)", B, T, F);
  DFA_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!ctx->mse_partial) DFA_HIP_CHECK(ctx, hipMalloc((void**)&ctx->mse_partial, kMseBlocks * sizeof(float)));
  DFA_HIP_CHECK(ctx, launch_mse_fwd_bwd(recon, x, x_dtype == DFA_DTYPE_BF16 ? 1 : 0, stride_b, stride_t, stride_f, B, T, F, ctx->mse_partial,
                                        loss, drecon, ctx->stream));
  return DFA_OK;
}

}  // extern "C"
