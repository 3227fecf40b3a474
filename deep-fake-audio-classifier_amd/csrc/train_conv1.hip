// train_conv1.hip -- train-mode passes over the 1-channel first block (src/model.py:15-19) that need the pre-BN
// convolution output z1 = conv1(x) + b.  z1 is never stored (7.4 MB/utt in fp32): with K = 9 it is cheaper to
// recompute it from x in each pass.  One kernel, three modes:
//   STATS      : per-channel sum / sum-of-squares of z1 over ALL T rows (BatchNorm sees the odd last row the pool drops)
//   BWD_REDUCE : S1 = sum dy, S2 = sum dy*xhat    with dy = relu'(y) * 0.5 * dropscale * da1[b][t/2][f][c]
//   WGRAD      : dz1 = gamma*invstd*(dy - S1/N - xhat*S2/N);  dW1[c][k] = sum dz1 * x_tap_k,  db1[c] = sum dz1
// Thread = (channel c, pixel lane): the 32 channel-threads of a pixel lane broadcast-read the same x taps from LDS
// and keep their channel's 9 weights and all accumulators in registers; a block covers an 8 x 64 pixel tile.
#include "dfa_internal.h"
#include "rng.h"

namespace dfa {

enum { C1M_STATS = 0, C1M_BWD_REDUCE = 1, C1M_WGRAD = 2 };
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

template <typename TX, typename T, int MODE>
__global__ __launch_bounds__(256) void conv1_train_kernel(const TX* __restrict__ x, int64_t sb, int64_t st, int64_t sf,
                                                          const float* __restrict__ w, const float* __restrict__ bconv,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ sums,
                                                          const T* __restrict__ da1, float* __restrict__ partial, int Tt,
                                                          int F, DropCfg dc, float inv_n) {
  constexpr int NV = (MODE == C1M_WGRAD) ? 10 : 2;
  __shared__ float xs[C1T_R + 2][C1T_C + 3];
  __shared__ float red[8][32][NV];
  const int tid = threadIdx.x, c = tid & 31, pl = tid >> 5;
  const int b = blockIdx.z, f0 = blockIdx.x * C1T_C;
  const TX* xb = x + (int64_t)b * sb;
  const bool t_fast = (st == 1);
  float wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = w[c * 9 + k];
  const float bc = bconv[c];
  float mu = 0.f, is = 0.f, gm = 0.f, bt = 0.f, s1n = 0.f, s2n = 0.f;
  if (MODE != C1M_STATS) { mu = mean[c]; is = invstd[c]; gm = gamma[c]; bt = beta[c]; }
  if (MODE == C1M_WGRAD) { s1n = sums[2 * c] * inv_n; s2n = sums[2 * c + 1] * inv_n; }
  float acc[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) acc[j] = 0.f;
  const int Ho = Tt >> 1;
  // the block walks its share of the 8-row tiles of this (utterance, 64-column strip): blockIdx.y, +gridDim.y, ...
  for (int t0 = blockIdx.y * C1T_R; t0 < Tt; t0 += gridDim.y * C1T_R) {
  __syncthreads();
  for (int e = tid; e < (C1T_R + 2) * (C1T_C + 2); e += 256) {
    int rr, cc;
    if (t_fast) { cc = e / (C1T_R + 2); rr = e - cc * (C1T_R + 2); } else { rr = e / (C1T_C + 2); cc = e - rr * (C1T_C + 2); }
    const int t = t0 - 1 + rr, f = f0 - 1 + cc;
    xs[rr][cc] = (t >= 0 && t < Tt && f >= 0 && f < F) ? ldx<TX>(xb + (int64_t)t * st + (int64_t)f * sf) : 0.f;
  }
  __syncthreads();
  for (int p = pl; p < C1T_R * C1T_C; p += 8) {
    const int rr = p / C1T_C, cc = p - rr * C1T_C;
    const int t = t0 + rr, f = f0 + cc;
    if (t >= Tt || f >= F) continue;
    float xv[9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) xv[dy * 3 + dx] = xs[rr + dy][cc + dx];
    float z = bc;
#pragma unroll
    for (int k = 0; k < 9; ++k) z = fmaf(wk[k], xv[k], z);
    if (MODE == C1M_STATS) {
      acc[0] += z;
      acc[1] = fmaf(z, z, acc[1]);
    } else {
      const float xh = (z - mu) * is;
      float dy = 0.f;
      const int to = t >> 1;
      if (to < Ho && fmaf(gm, xh, bt) > 0.f) {
        const size_t idx = (((size_t)b * Ho + to) * F + f) * 32 + c;
        float g = 0.5f * ldf<T>(da1 + idx);
        if (dc.thresh != 0) {
          float ds[8];
          drop_scale8(dc, idx & ~(size_t)7, ds);
          g *= ds[c & 7];
        }
        dy = g;
      }
      if (MODE == C1M_BWD_REDUCE) {
        acc[0] += dy;
        acc[1] = fmaf(dy, xh, acc[1]);
      } else {
        const float dz = gm * is * (dy - s1n - xh * s2n);
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = fmaf(dz, xv[k], acc[k]);
        acc[9] += dz;
      }
    }
  }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j) red[pl][c][j] = acc[j];
  __syncthreads();
  const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  for (int e = tid; e < 32 * NV; e += 256) {
    const int cc = e / NV, j = e - cc * NV;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][cc][j];
    partial[blk * (32 * NV) + e] = s;
  }
}

constexpr int C1T_GY = 4;  // row-tile walkers per (utterance, column strip)
int conv1_train_blocks(int B, int T, int F) { (void)T; return B * C1T_GY * ((F + C1T_C - 1) / C1T_C); }

hipError_t launch_conv1_train(int mode, const void* x, int x_dtype, int64_t sb, int64_t st, int64_t sf, const float* w,
                              const float* bconv, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, const float* sums, const void* da1, int prec, float* partial, int B,
                              int T, int F, const DropCfg& dc, hipStream_t s) {
  dim3 grid((F + C1T_C - 1) / C1T_C, C1T_GY, B), block(256);
  const float inv_n = (float)(1.0 / ((double)B * T * F));
#define DFA_C1T(TXX, TT, MODE)                                                                                        \
  hipLaunchKernelGGL((conv1_train_kernel<TXX, TT, MODE>), grid, block, 0, s, (const TXX*)x, sb, st, sf, w, bconv, mean, \
                     invstd, gamma, beta, sums, (const TT*)da1, partial, T, F, dc, inv_n)
#define DFA_C1T_MODES(TXX, TT)                                                                                        \
  do {                                                                                                                 \
    if (mode == C1M_STATS) DFA_C1T(TXX, TT, C1M_STATS);                                                                \
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
