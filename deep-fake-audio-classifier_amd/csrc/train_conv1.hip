// train_conv1.hip -- train-mode passes over the 1-channel first block (src/model.py:15-19) that need the pre-BN
// convolution output z1 = conv1(x) + b.  z1 is never stored (7.4 MB/utt in fp32): with K = 9 it is cheaper to
// recompute it from x in each pass.  One kernel, three modes:
//   STATS      : per-channel sum / sum-of-squares of z1 over ALL T rows (BatchNorm sees the odd last row the pool drops)
//   BWD_REDUCE : S1 = sum dy, S2 = sum dy*xhat    with dy = relu'(y) * 0.5 * dropscale * da1[b][t/2][f][c]
//   WGRAD      : dz1 = gamma*invstd*(dy - S1/N - xhat*S2/N);  dW1[c][k] = sum dz1 * x_tap_k,  db1[c] = sum dz1
//   BWD_FUSED  : BWD_REDUCE and WGRAD in ONE pass over da1 (round 2).  dz1 is linear in (S1, S2), so
//                  dW1[c][k] = ga_c * ( A[c][k] - S1_c/N * Xs[k] - S2_c/N * Hx[c][k] ),   A[c][k] = sum dy * x_k,
//                  Xs[k] = sum x_k,   Hx[c][k] = sum xhat_c * x_k = is_c * (b_c*Xs[k] + sum_j w_c[j]*XX[j][k]) - mu_c*is_c*Xs[k],
//                with XX[j][k] = sum x_j * x_k a 9 x 9 channel-independent matrix that the STATS pass of the forward
//                accumulates on the side (STATS_XX).  The pass accumulates A, S1, S2; conv1_bwd_finalize_kernel does the algebra.
// Thread = (channel c, pixel lane): the 32 channel-threads of a pixel lane broadcast-read the same x taps from LDS
// and keep their channel's 9 weights and all accumulators in registers; a block covers an 8 x 64 pixel tile.
#include "dfa_internal.h"
#include "rng.h"

namespace dfa {

enum { C1M_STATS = 0, C1M_BWD_REDUCE = 1, C1M_WGRAD = 2, C1M_BWD_FUSED = 3, C1M_STATS_XX = 4 };
constexpr int C1T_R = 8, C1T_C = 64;

template <typename TX>
__device__ __forceinline__ float ldx(const TX* p);
template <>
__device__ __forceinline__ float ldx<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldx<bf16_t>(const bf16_t* p) { return bf16_to_float(*p); }
template <typename T>
__device__ __forceinline__ float ldf(const T* p);
template <>
__device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldf<bf16_t>(const bf16_t* p) { return bf16_to_float(*p); }

template <typename T>
__device__ __forceinline__ void ld8f(const T* p, float* v);
template <>
__device__ __forceinline__ void ld8f<float>(const float* p, float* v) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void ld8f<bf16_t>(const bf16_t* p, float* v) {
  const uint4 q = *reinterpret_cast<const uint4*>(p);
  const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(u[e] << 16); v[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
}

// Thread = (pixel lane, channel octet): lanes 4p..4p+3 own the four 8-channel groups of a pixel, so the upstream
// gradient da1 is one 16/32-byte load per thread, the dropout mask costs two Philox calls per 8 elements, and the x taps
// are LDS broadcasts.  Every thread keeps its 8 channels' weights, BN constants and accumulators in registers and walks
// 8 pixels of each 8 x 64 tile; sums are combined across the 64 pixel lanes at the end (shuffles, then LDS).
template <typename TX, typename T, int MODE, int POOLW = 1>
__global__ __launch_bounds__(256) void conv1_train_kernel(const TX* __restrict__ x, int64_t sb, int64_t st, int64_t sf,
                                                          const float* __restrict__ w, const float* __restrict__ bconv,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ sums,
                                                          const T* __restrict__ da1, float* __restrict__ partial, int Tt,
                                                          int F, DropCfg dc, float inv_n, AugCfg aug) {
  constexpr int NV = (MODE == C1M_WGRAD) ? 10 : (MODE == C1M_BWD_FUSED) ? 11 : 2;
  constexpr bool STATS = (MODE == C1M_STATS || MODE == C1M_STATS_XX);
  // STATS_XX: thread q also owns rows j = q, q+4, q+8 of XX[j][k] = sum x_j*x_k, and thread q == 1 the tap sums Xs[k]
  float xx[MODE == C1M_STATS_XX ? 3 : 1][9], xsum[9];
  int jr[3], jc[3];                                  // tap j = q + 4*jj as (row, column) offsets into the x tile
#pragma unroll
  for (int jj = 0; jj < 3; ++jj) { const int jt = min((threadIdx.x & 3) + 4 * jj, 8); jr[jj] = jt / 3; jc[jj] = jt - 3 * jr[jj]; }
#pragma unroll
  for (int j = 0; j < (MODE == C1M_STATS_XX ? 3 : 1); ++j)
#pragma unroll
    for (int k = 0; k < 9; ++k) xx[j][k] = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) xsum[k] = 0.f;
  __shared__ float xs[C1T_R + 2][C1T_C + 3];
  __shared__ float red[4][4][(8 * NV) < 24 ? 24 : 8 * NV];   // >= 4 x 96 floats for the STATS_XX block record
  const int tid = threadIdx.x, q = tid & 3, pl = tid >> 2;
  const int b = blockIdx.z, f0 = blockIdx.x * C1T_C;
  const TX* xb = x + (int64_t)b * sb;
  const bool t_fast = (st == 1);
  float wk[8][9], bc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[c][k] = w[(q * 8 + c) * 9 + k];
    bc[c] = bconv[q * 8 + c];
  }
  // BN constants folded per channel: xhat = z*is - mu*is;  y > 0 test on gm*xhat + bt;  dz = ga*(dy - s1n - xhat*s2n)
  float is[8], mis[8], gm[8], bt[8], ga[8], s1n[8], s2n[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int ch = q * 8 + c;
    is[c] = mis[c] = gm[c] = bt[c] = ga[c] = s1n[c] = s2n[c] = 0.f;
    if (!STATS) { is[c] = invstd[ch]; mis[c] = mean[ch] * is[c]; gm[c] = gamma[ch]; bt[c] = beta[ch]; }
    if (MODE == C1M_WGRAD) { ga[c] = gm[c] * is[c]; s1n[c] = sums[2 * ch] * inv_n; s2n[c] = sums[2 * ch + 1] * inv_n; }
  }
  float acc[8][NV];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[c][j] = 0.f;
  const int Ho = Tt >> 1;
  // the block walks its share of the 8-row tiles of this (utterance, 64-column strip): blockIdx.y, +gridDim.y, ...
  for (int t0 = blockIdx.y * C1T_R; t0 < Tt; t0 += gridDim.y * C1T_R) {
    __syncthreads();
    for (int e = tid; e < (C1T_R + 2) * (C1T_C + 2); e += 256) {
      int rr, cc;
      if (t_fast) { cc = e / (C1T_R + 2); rr = e - cc * (C1T_R + 2); } else { rr = e / (C1T_C + 2); cc = e - rr * (C1T_C + 2); }
      const int t = t0 - 1 + rr, f = f0 - 1 + cc;
      xs[rr][cc] = (t >= 0 && t < Tt && f >= 0 && f < F)
                       ? aug_apply(aug, ldx<TX>(xb + (int64_t)aug_src_t(aug, t) * st + (int64_t)f * sf), b, t, f) : 0.f;
    }
    __syncthreads();
#pragma unroll 1
    for (int rr = 0; rr < C1T_R; ++rr) {
      const int t = t0 + rr, f = f0 + pl;
      if (t >= Tt || f >= F) continue;
      float xv[9];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) xv[dy * 3 + dx] = xs[rr + dy][pl + dx];
      float g[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) g[c] = 0.f;
      const int to = t >> 1;
      if (MODE == C1M_STATS_XX) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const float xj = xs[rr + jr[j]][pl + jc[j]];      // rows j >= 9 (q = 1..3, jj = 2) are computed and never stored
#pragma unroll
          for (int k = 0; k < 9; ++k) xx[j][k] = fmaf(xj, xv[k], xx[j][k]);
        }
        if (q == 1) {
#pragma unroll
          for (int k = 0; k < 9; ++k) xsum[k] += xv[k];
        }
      }
      if (!STATS && POOLW == 1 && to < Ho) {          // AvgPool2d((2,1)) + Dropout upstream (CNN2D)
        const size_t idx = (((size_t)b * Ho + to) * F + f) * 32 + q * 8;
        float d[8], ds[8];
        ld8f<T>(da1 + idx, d);
        drop_scale8(dc, idx, ds);
#pragma unroll
        for (int c = 0; c < 8; ++c) g[c] = 0.5f * d[c] * ds[c];
      }
      if (!STATS && POOLW == 2 && to < Ho && (f >> 1) < (F >> 1)) {   // AvgPool2d(2) upstream (CAE)
        const size_t idx = (((size_t)b * Ho + to) * (F >> 1) + (f >> 1)) * 32 + q * 8;
        float d[8];
        ld8f<T>(da1 + idx, d);
#pragma unroll
        for (int c = 0; c < 8; ++c) g[c] = 0.25f * d[c];
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        float z = bc[c];
#pragma unroll
        for (int k = 0; k < 9; ++k) z = fmaf(wk[c][k], xv[k], z);
        if (STATS) {
          acc[c][0] += z;
          acc[c][1] = fmaf(z, z, acc[c][1]);
        } else {
          const float xh = fmaf(z, is[c], -mis[c]);
          const float dy = (fmaf(gm[c], xh, bt[c]) > 0.f) ? g[c] : 0.f;
          if (MODE == C1M_BWD_REDUCE) {
            acc[c][0] += dy;
            acc[c][1] = fmaf(dy, xh, acc[c][1]);
          } else if (MODE == C1M_BWD_FUSED) {
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[c][k] = fmaf(dy, xv[k], acc[c][k]);
            acc[c][9] += dy;
            acc[c][10] = fmaf(dy, xh, acc[c][10]);
          } else {
            const float dz = ga[c] * (dy - s1n[c] - xh * s2n[c]);
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[c][k] = fmaf(dz, xv[k], acc[c][k]);
            acc[c][NV - 1] += dz;
          }
        }
      }
    }
  }
  // combine the 16 pixel lanes of each wave that share an octet (lane bits 2..5), then the 4 waves through LDS
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      float v = acc[c][j];
#pragma unroll
      for (int off = 4; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
      acc[c][j] = v;
    }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane < 4) {
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int j = 0; j < NV; ++j) red[wave][lane][c * NV + j] = acc[c][j];
  }
  __syncthreads();
  const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  for (int e = tid; e < 32 * NV; e += 256) {   // e = channel * NV + j, channel = octet * 8 + c
    const int ch = e / NV, j = e - ch * NV;
    const int qq = ch >> 3, c = ch & 7;
    partial[blk * (32 * NV) + e] = (red[0][qq][c * NV + j] + red[1][qq][c * NV + j]) +
                                   (red[2][qq][c * NV + j] + red[3][qq][c * NV + j]);
  }
  if (MODE == C1M_STATS_XX) {
    // second record per block, behind all the [32][2] records: XX[9][9] then Xs[9] (same lane-then-wave combination)
    __syncthreads();
    float* red2 = &red[0][0][0];                     // reuse: [4 waves][90]
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        float v = xx[j][k];
#pragma unroll
        for (int off = 4; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
        if (lane < 4 && lane + 4 * j < 9) red2[wave * 96 + (lane + 4 * j) * 9 + k] = v;
      }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      float v = xsum[k];
#pragma unroll
      for (int off = 4; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
      if (lane == 1) red2[wave * 96 + 81 + k] = v;
    }
    __syncthreads();
    float* part2 = partial + (size_t)gridDim.x * gridDim.y * gridDim.z * (32 * NV);
    if (tid < 90) part2[blk * 96 + tid] = (red2[tid] + red2[96 + tid]) + (red2[192 + tid] + red2[288 + tid]);
  }
}

// dW1, db1, dgamma1, dbeta1 from the fused backward record rec[32][11] (A[9], S1, S2), the forward's xxs[90] (XX[9][9], Xs[9])
// and the layer's constants -- the algebra in the header comment, in double.
__global__ void conv1_bwd_finalize_kernel(const float* __restrict__ rec, const float* __restrict__ xxs,
                                          const float* __restrict__ w, const float* __restrict__ bconv,
                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                          const float* __restrict__ gamma, double n, float* __restrict__ dw,
                                          float* __restrict__ db, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                          int derive_s2) {
  const int i = threadIdx.x;          // 320 threads: (channel c, k) with k = 9 -> the bias / BN entries
  const int c = i / 10, k = i - c * 10;
  if (c >= 32) return;
  const double is = invstd[c], mu = mean[c], ga = (double)gamma[c] * is;
  const double S1 = rec[c * 11 + 9];
  double S2 = rec[c * 11 + 10];
  if (derive_s2) {   // train_conv1_mfma.hip: sum dy*xhat = is * (sum dy*z - mu*S1),  sum dy*z = b*S1 + sum_j w_j * A[c][j]
    double dz = (double)bconv[c] * S1;
    for (int j = 0; j < 9; ++j) dz += (double)w[c * 9 + j] * (double)rec[c * 11 + j];
    S2 = is * (dz - mu * S1);
  }
  if (k < 9) {
    const double Xs = xxs[81 + k];
    double zx = (double)bconv[c] * Xs;
    for (int j = 0; j < 9; ++j) zx += (double)w[c * 9 + j] * (double)xxs[j * 9 + k];
    const double hx = is * zx - mu * is * Xs;
    dw[c * 9 + k] = (float)(ga * ((double)rec[c * 11 + k] - S1 / n * Xs - S2 / n * hx));
  } else {
    // db = sum dz = ga * (S1 - S1 - S2/N * sum xhat); sum xhat = is * (sum z - N*mu), sum z = N*b + sum_j w_j * Xs[j]
    double sz = n * (double)bconv[c];
    for (int j = 0; j < 9; ++j) sz += (double)w[c * 9 + j] * (double)xxs[81 + j];
    db[c] = (float)(-ga * S2 / n * (is * (sz - n * mu)));
    dgamma[c] = (float)S2;
    dbeta[c] = (float)S1;
  }
}

hipError_t launch_conv1_bwd_finalize(const float* rec, const float* xxs, const float* w, const float* bconv, const float* mean,
                                     const float* invstd, const float* gamma, double n, float* dw, float* db, float* dgamma,
                                     float* dbeta, hipStream_t s, int derive_s2) {
  hipLaunchKernelGGL(conv1_bwd_finalize_kernel, dim3(1), dim3(320), 0, s, rec, xxs, w, bconv, mean, invstd, gamma, n, dw, db,
                     dgamma, dbeta, derive_s2);
  return hipGetLastError();
}

constexpr int C1T_GY = 4;  // row-tile walkers per (utterance, column strip)
int conv1_train_blocks(int B, int T, int F) { (void)T; return B * C1T_GY * ((F + C1T_C - 1) / C1T_C); }

hipError_t launch_conv1_train(int mode, const void* x, int x_dtype, int64_t sb, int64_t st, int64_t sf, const float* w,
                              const float* bconv, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, const float* sums, const void* da1, int prec, float* partial, int B,
                              int T, int F, const DropCfg& dc, hipStream_t s, int poolw, const AugCfg* augp, float inv_n_scale) {
  AugCfg aug{};
  if (augp) aug = *augp;
  dim3 grid((F + C1T_C - 1) / C1T_C, C1T_GY, B), block(256);
  const float inv_n = (float)(1.0 / ((double)B * T * F)) * inv_n_scale;     // (synchronised BatchNorm: 1 / world -> 1 / n_global)
#define DFA_C1T(TXX, TT, MODE)                                                                                        \
  do {                                                                                                                 \
    if (poolw == 2)                                                                                                    \
      hipLaunchKernelGGL((conv1_train_kernel<TXX, TT, MODE, 2>), grid, block, 0, s, (const TXX*)x, sb, st, sf, w, bconv, \
                         mean, invstd, gamma, beta, sums, (const TT*)da1, partial, T, F, dc, inv_n, aug);                  \
    else                                                                                                               \
      hipLaunchKernelGGL((conv1_train_kernel<TXX, TT, MODE, 1>), grid, block, 0, s, (const TXX*)x, sb, st, sf, w, bconv, \
                         mean, invstd, gamma, beta, sums, (const TT*)da1, partial, T, F, dc, inv_n, aug);                  \
  } while (0)
#define DFA_C1T_MODES(TXX, TT)                                                                                        \
  do {                                                                                                                 \
    if (mode == C1M_STATS) DFA_C1T(TXX, TT, C1M_STATS);                                                                \
    else if (mode == C1M_STATS_XX) DFA_C1T(TXX, TT, C1M_STATS_XX);                                                     \
    else if (mode == C1M_BWD_FUSED) DFA_C1T(TXX, TT, C1M_BWD_FUSED);                                                   \
    else if (mode == C1M_BWD_REDUCE) DFA_C1T(TXX, TT, C1M_BWD_REDUCE);                                                 \
    else DFA_C1T(TXX, TT, C1M_WGRAD);                                                                                  \
  } while (0)
  if (x_dtype == DFA_DTYPE_F32 && prec == DFA_PREC_F32) DFA_C1T_MODES(float, float);
  else if (x_dtype == DFA_DTYPE_F32) DFA_C1T_MODES(float, bf16_t);
  else if (prec == DFA_PREC_F32) DFA_C1T_MODES(bf16_t, float);
  else DFA_C1T_MODES(bf16_t, bf16_t);
#undef DFA_C1T_MODES
#undef DFA_C1T
  return hipGetLastError();
}

}  // namespace dfa
