// train_cnn1d.hip -- training step of the 1-D CNN (src/train.py:71-76 over src/model_cnn1d.py:15-46): train-mode
// forward (Conv1d -> BatchNorm1d(batch statistics) -> ReLU -> Dropout, x3; mean over T; Linear) and its backward.
// Everything is channel-major [B][C][T] fp32 (the stored feature layout).  The whole network is 30.8 MFLOP/utt
// (about 0.3 % of the 2-D CNN): these are LDS-tiled fp32 VALU kernels bound by reading the 231 KB input, not matrix-core
// kernels.  Reductions are two-stage with a fixed order (deterministic).
#include "dfa_internal.h"
#include "rng.h"

namespace dfa {

// ---- per-channel sum / sum of squares of z[B][C][T]: one block per (channel, batch chunk) -> partial[chunk][C][2]
__global__ __launch_bounds__(256) void cm_stats_kernel(const float* __restrict__ z, float* __restrict__ partial, int B,
                                                       int C, int T, int bchunk) {
  __shared__ float r1[256], r2[256];
  const int c = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
  const int b0 = ch * bchunk, b1 = min(B, b0 + bchunk);
  float s1 = 0.f, s2 = 0.f;
  for (int b = b0; b < b1; ++b) {
    const float* row = z + ((size_t)b * C + c) * T;
    for (int t = tid; t < T; t += 256) { const float v = row[t]; s1 += v; s2 = fmaf(v, v, s2); }
  }
  r1[tid] = s1; r2[tid] = s2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { r1[tid] += r1[tid + off]; r2[tid] += r2[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) { partial[((size_t)ch * C + c) * 2] = r1[0]; partial[((size_t)ch * C + c) * 2 + 1] = r2[0]; }
}

__device__ __forceinline__ float drop1(const DropCfg& dc, uint64_t idx) {
  if (dc.thresh == 0) return 1.f;
  float f[8];
  drop_scale8(dc, idx & ~(uint64_t)7, f);
  return f[idx & 7];
}

// ---- forward: h = dropout(relu(bn(z)))  (elementwise, [B][C][T])
__global__ void cm_bn_relu_drop_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, float* __restrict__ h, int C, int T, size_t n,
                                       DropCfg dc) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)((i / T) % C);
  const float y = fmaf((z[i] - mean[c]) * invstd[c], gamma[c], beta[c]);
  h[i] = fmaxf(y, 0.f) * drop1(dc, i);
}

// ---- forward: pooled[b][c] = mean_t relu(bn(z[b][c][t]))   (one wave per (b, c) row)
__global__ __launch_bounds__(256) void cm_bn_relu_meant_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ pooled,
                                                               int C, int T, int rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const int c = row % C;
  const float sc = gamma[c] * invstd[c], sh = beta[c] - mean[c] * sc;
  float s = 0.f;
  for (int t = lane; t < T; t += 64) s += fmaxf(fmaf(z[(size_t)row * T + t], sc, sh), 0.f);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) pooled[row] = s / (float)T;
}

// ---- BatchNorm1d backward, channel-major.  Upstream gradient of the BN output after the ReLU mask:
//   SRC 0 (mean over T then Linear): dy = (y > 0) * dpooled[b][c] / T;   SRC 1 (dropout): dy = (y > 0) * dropscale * dh[b][c][t]
template <int SRC>
__global__ __launch_bounds__(256) void cm_bn_bwd_reduce_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ up, float* __restrict__ partial,
                                                               int B, int C, int T, int bchunk, DropCfg dc) {
  __shared__ float r1[256], r2[256];
  const int c = blockIdx.x, ch = blockIdx.y, tid = threadIdx.x;
  const int b0 = ch * bchunk, b1 = min(B, b0 + bchunk);
  const float mu = mean[c], is = invstd[c], gm = gamma[c], bt = beta[c];
  float s1 = 0.f, s2 = 0.f;
  for (int b = b0; b < b1; ++b) {
    const size_t base = ((size_t)b * C + c) * T;
    for (int t = tid; t < T; t += 256) {
      const float xh = (z[base + t] - mu) * is;
      float g = (SRC == 0) ? up[(size_t)b * C + c] / (float)T : up[base + t] * drop1(dc, base + t);
      const float dy = (fmaf(gm, xh, bt) > 0.f) ? g : 0.f;
      s1 += dy;
      s2 = fmaf(dy, xh, s2);
    }
  }
  r1[tid] = s1; r2[tid] = s2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { r1[tid] += r1[tid + off]; r2[tid] += r2[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) { partial[((size_t)ch * C + c) * 2] = r1[0]; partial[((size_t)ch * C + c) * 2 + 1] = r2[0]; }
}

template <int SRC>
__global__ void cm_bn_bwd_apply_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                       const float* __restrict__ beta, const float* __restrict__ sums,
                                       const float* __restrict__ up, float* __restrict__ dz, int C, int T, size_t n,
                                       float inv_n, DropCfg dc) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)((i / T) % C);
  const size_t bc = i / T;
  const float xh = (z[i] - mean[c]) * invstd[c];
  const float g = (SRC == 0) ? up[bc] / (float)T : up[i] * drop1(dc, i);
  const float dy = (fmaf(gamma[c], xh, beta[c]) > 0.f) ? g : 0.f;
  dz[i] = gamma[c] * invstd[c] * (dy - sums[2 * c] * inv_n - xh * sums[2 * c + 1] * inv_n);
}

// ---- Conv1d weight gradient: dW[o][c][k] = sum_{b,t} dz[b][o][t] * h[b][c][t+k-1],  db[o] = sum dz.
// Block = (16 output channels x 16 input channels) tile for one batch chunk; thread (o, c) keeps its 3 taps (+ bias
// sum) in registers and walks time through LDS slabs of 64 frames.  partial[chunk][Cout][Cin][3] (+ [Cout] for db).
constexpr int W1D_TT = 64;
__global__ __launch_bounds__(256) void conv1d_wgrad_kernel(const float* __restrict__ dz, const float* __restrict__ h,
                                                           int64_t hsb, int64_t hsc, int64_t hst,
                                                           float* __restrict__ partial, int B, int Cin, int Cout, int T,
                                                           int bchunk) {
  __shared__ float dzs[16][W1D_TT];
  __shared__ float hs[16][W1D_TT + 2];
  const int tid = threadIdx.x, ol = tid >> 4, cl = tid & 15;
  const int o0 = blockIdx.x * 16, c0 = blockIdx.y * 16, ch = blockIdx.z;
  const int b0 = ch * bchunk, b1 = min(B, b0 + bchunk);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, ab = 0.f;
  for (int b = b0; b < b1; ++b) {
    for (int t0 = 0; t0 < T; t0 += W1D_TT) {
      __syncthreads();
      for (int e = tid; e < 16 * W1D_TT; e += 256) {
        const int oo = e / W1D_TT, tt = e - oo * W1D_TT;
        dzs[oo][tt] = (t0 + tt < T) ? dz[((size_t)b * Cout + o0 + oo) * T + t0 + tt] : 0.f;
      }
      for (int e = tid; e < 16 * (W1D_TT + 2); e += 256) {
        const int cc = e / (W1D_TT + 2), tt = e - cc * (W1D_TT + 2);
        const int t = t0 - 1 + tt, ci = c0 + cc;
        hs[cc][tt] = (ci < Cin && t >= 0 && t < T) ? h[(int64_t)b * hsb + (int64_t)ci * hsc + (int64_t)t * hst] : 0.f;
      }
      __syncthreads();
#pragma unroll 8
      for (int tt = 0; tt < W1D_TT; ++tt) {
        const float d = dzs[ol][tt];
        a0 = fmaf(d, hs[cl][tt], a0);
        a1 = fmaf(d, hs[cl][tt + 1], a1);
        a2 = fmaf(d, hs[cl][tt + 2], a2);
        ab += d;
      }
    }
  }
  const int o = o0 + ol, c = c0 + cl;
  float* rec = partial + (size_t)ch * ((size_t)Cout * Cin * 3 + Cout);
  if (c < Cin) {
    rec[((size_t)o * Cin + c) * 3] = a0;
    rec[((size_t)o * Cin + c) * 3 + 1] = a1;
    rec[((size_t)o * Cin + c) * 3 + 2] = a2;
  }
  if (blockIdx.y == 0 && cl == 0) rec[(size_t)Cout * Cin * 3 + o] = ab;
}

// ---- the same weight gradient on the fp32 matrix cores: three GEMMs [32 o x t] . [t x 32 c] (one per tap) sharing the dz
// operand; both operands are channel-major with t contiguous, i.e. K-contiguous rows, so v_mfma_f32_32x32x2_f32 takes
// them straight from LDS tiles (lane = channel, k = frame parity).  One wave per (o tile, c tile, utterance chunk):
// 96 MFMAs per 64-frame slab; the VALU kernel above spent 4 LDS reads on every 3 FMAs and ran at 128-512 blocks.
// AUG (layer 1 only): h = x read through the armed train-time augmentation, as the forward read it (conv1d.hip)
template <bool AUG>
__global__ __launch_bounds__(64) void conv1d_wgrad_mfma_kernel(const float* __restrict__ dz, const float* __restrict__ h,
                                                               int64_t hsb, int64_t hsc, int64_t hst,
                                                               float* __restrict__ partial, int B, int Cin, int Cout, int T,
                                                               int bchunk, AugCfg aug) {
  __shared__ float dzs[32][W1D_TT + 1];
  __shared__ float hs[32][W1D_TT + 3];
  const int lane = threadIdx.x, r = lane & 31, hh = lane >> 5;
  const int o0 = blockIdx.x * 32, c0 = blockIdx.y * 32, ch = blockIdx.z;
  const int b0 = ch * bchunk, b1 = min(B, b0 + bchunk);
  f32x16_t acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  float ab = 0.f;
  // slab = (utterance, 64 frames); the next slab's 66 values per lane are fetched into registers while the MFMAs of the
  // current one run (all loads of a slab are issued back to back: one latency per slab, not one per row)
  const int nslab_t = (T + W1D_TT - 1) / W1D_TT;
  const int nslab = (b1 - b0) * nslab_t;
  float rdz[32], rh[32], rh2 = 0.f;
  auto fetch = [&](int sl) {
    const int b = b0 + sl / nslab_t, t0 = (sl % nslab_t) * W1D_TT;
#pragma unroll
    for (int oo = 0; oo < 32; ++oo)
      rdz[oo] = (t0 + lane < T) ? dz[((size_t)b * Cout + o0 + oo) * T + t0 + lane] : 0.f;
    const int t = t0 - 1 + lane;
#pragma unroll
    for (int cc = 0; cc < 32; ++cc) {
      const int ci = c0 + cc;
      if constexpr (AUG)
        rh[cc] = (ci < Cin && t >= 0 && t < T) ? aug_apply(aug, h[(int64_t)b * hsb + (int64_t)ci * hsc + (int64_t)aug_src_t(aug, t) * hst], b, t, ci) : 0.f;
      else
        rh[cc] = (ci < Cin && t >= 0 && t < T) ? h[(int64_t)b * hsb + (int64_t)ci * hsc + (int64_t)t * hst] : 0.f;
    }
    // the two extra frames of the 66-frame window: lane = (channel, which frame)
    const int ci2 = c0 + (lane & 31), t2 = t0 + 63 + (lane >> 5);
    if constexpr (AUG)
      rh2 = (ci2 < Cin && t2 < T) ? aug_apply(aug, h[(int64_t)b * hsb + (int64_t)ci2 * hsc + (int64_t)aug_src_t(aug, t2) * hst], b, t2, ci2) : 0.f;
    else
      rh2 = (ci2 < Cin && t2 < T) ? h[(int64_t)b * hsb + (int64_t)ci2 * hsc + (int64_t)t2 * hst] : 0.f;
  };
  if (nslab > 0) fetch(0);
  for (int sl = 0; sl < nslab; ++sl) {
    __syncthreads();
#pragma unroll
    for (int oo = 0; oo < 32; ++oo) dzs[oo][lane] = rdz[oo];
#pragma unroll
    for (int cc = 0; cc < 32; ++cc) hs[cc][lane] = rh[cc];
    hs[lane & 31][64 + (lane >> 5)] = rh2;
    __syncthreads();
    if (sl + 1 < nslab) fetch(sl + 1);
#pragma unroll 8
    for (int tt = 0; tt < W1D_TT; tt += 2) {
      const float a = dzs[r][tt + hh];
      ab += a;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, hs[r][tt + hh + k], acc[k], 0, 0, 0);
    }
  }
  float* rec = partial + (size_t)ch * ((size_t)Cout * Cin * 3 + Cout);
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int o = o0 + (i & 3) + 8 * (i >> 2) + 4 * hh, c = c0 + r;
      if (c < Cin) rec[((size_t)o * Cin + c) * 3 + k] = acc[k][i];
    }
  ab += __shfl_xor(ab, 32, 64);
  if (blockIdx.y == 0 && hh == 0) rec[(size_t)Cout * Cin * 3 + o0 + r] = ab;
}

// ---- the same weight gradient on the bf16 matrix cores at fp32 grade: every fp32 operand is carried as three bf16 terms
// (hi + lo + lo2 = its 24-bit mantissa exactly) and every product is six v_mfma_f32_32x32x16_bf16 (all term pairs of order <= 2),
// as the training convolutions of cnn1d_fused_x3.hip do.  The fp32 matrix pipe issues one 32x32x2 MFMA per ~80 cycles: 24 of them
// per 16 frames and tap triple = 1920 cycles; the same products cost 18 x 32 = 576 matrix-pipe cycles here plus the splits.
// Same decomposition and slab pipeline as the kernel above; the LDS tiles are laid out for aligned 16-byte fragment reads (row
// pitch 68 floats: the 16 lanes of a ds_read_b128 group fall on distinct banks): a lane reads its row's 8 dz frames (two reads)
// and 12 h frames (three reads: frames t-1 .. t+10) per 16-frame k-step, the three taps are register windows [k, k + 8) of
// those 12 values.
__device__ __forceinline__ void w1d_split3(const float (&v)[8], uint4 (&f)[3]) {
  unsigned q[3][4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float r0 = v[2 * p], r1 = v[2 * p + 1];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      q[t][p] = pack_bf16x2(r0, r1);
      if (t < 2) { r0 -= __uint_as_float(q[t][p] << 16); r1 -= __uint_as_float(q[t][p] & 0xffff0000u); }
    }
  }
#pragma unroll
  for (int t = 0; t < 3; ++t) f[t] = make_uint4(q[t][0], q[t][1], q[t][2], q[t][3]);
}
__device__ __forceinline__ f32x16_t w1d_mma(const uint4& a, const uint4& b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

template <bool AUG>
__global__ __launch_bounds__(64) void conv1d_wgrad_x3_kernel(const float* __restrict__ dz, const float* __restrict__ h,
                                                             int64_t hsb, int64_t hsc, int64_t hst,
                                                             float* __restrict__ partial, int B, int Cin, int Cout, int T,
                                                             int bchunk, AugCfg aug) {
  constexpr int PITCH = 68;
  __shared__ __attribute__((aligned(16))) float dzs[32][PITCH];
  __shared__ __attribute__((aligned(16))) float hs[32][PITCH];      // hs[c][i] = frame t0 - 1 + i, i < 66 (67 = zero)
  const int lane = threadIdx.x, r = lane & 31, hh = lane >> 5;
  const int o0 = blockIdx.x * 32, c0 = blockIdx.y * 32, ch = blockIdx.z;
  const int b0 = ch * bchunk, b1 = min(B, b0 + bchunk);
  f32x16_t acc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  float ab = 0.f;
  const int nslab_t = (T + W1D_TT - 1) / W1D_TT;
  const int nslab = (b1 - b0) * nslab_t;
  float rdz[32], rh[32], rh2 = 0.f;
  auto fetch = [&](int sl) {
    const int b = b0 + sl / nslab_t, t0 = (sl % nslab_t) * W1D_TT;
#pragma unroll
    for (int oo = 0; oo < 32; ++oo)
      rdz[oo] = (t0 + lane < T) ? dz[((size_t)b * Cout + o0 + oo) * T + t0 + lane] : 0.f;
    const int t = t0 - 1 + lane;
#pragma unroll
    for (int cc = 0; cc < 32; ++cc) {
      const int ci = c0 + cc;
      if constexpr (AUG)
        rh[cc] = (ci < Cin && t >= 0 && t < T) ? aug_apply(aug, h[(int64_t)b * hsb + (int64_t)ci * hsc + (int64_t)aug_src_t(aug, t) * hst], b, t, ci) : 0.f;
      else
        rh[cc] = (ci < Cin && t >= 0 && t < T) ? h[(int64_t)b * hsb + (int64_t)ci * hsc + (int64_t)t * hst] : 0.f;
    }
    const int ci2 = c0 + (lane & 31), t2 = t0 + 63 + (lane >> 5);
    if constexpr (AUG)
      rh2 = (ci2 < Cin && t2 < T) ? aug_apply(aug, h[(int64_t)b * hsb + (int64_t)ci2 * hsc + (int64_t)aug_src_t(aug, t2) * hst], b, t2, ci2) : 0.f;
    else
      rh2 = (ci2 < Cin && t2 < T) ? h[(int64_t)b * hsb + (int64_t)ci2 * hsc + (int64_t)t2 * hst] : 0.f;
  };
  if (lane < 32) { hs[lane][66] = 0.f; hs[lane][67] = 0.f; }      // the tail of the 12-frame window of the last k-step
  if (nslab > 0) fetch(0);
  for (int sl = 0; sl < nslab; ++sl) {
    __syncthreads();
#pragma unroll
    for (int oo = 0; oo < 32; ++oo) dzs[oo][lane] = rdz[oo];
#pragma unroll
    for (int cc = 0; cc < 32; ++cc) hs[cc][lane] = rh[cc];
    hs[lane & 31][64 + (lane >> 5)] = rh2;
    __syncthreads();
    if (sl + 1 < nslab) fetch(sl + 1);
#pragma unroll
    for (int ks = 0; ks < W1D_TT / 16; ++ks) {
      float dv[8], hv[12];
      *(float4*)&dv[0] = *(const float4*)&dzs[r][16 * ks + 8 * hh];
      *(float4*)&dv[4] = *(const float4*)&dzs[r][16 * ks + 8 * hh + 4];
      *(float4*)&hv[0] = *(const float4*)&hs[r][16 * ks + 8 * hh];
      *(float4*)&hv[4] = *(const float4*)&hs[r][16 * ks + 8 * hh + 4];
      *(float4*)&hv[8] = *(const float4*)&hs[r][16 * ks + 8 * hh + 8];
#pragma unroll
      for (int j = 0; j < 8; ++j) ab += dv[j];
      uint4 a[3];
      w1d_split3(dv, a);
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float win[8] = {hv[k], hv[k + 1], hv[k + 2], hv[k + 3], hv[k + 4], hv[k + 5], hv[k + 6], hv[k + 7]};
        uint4 bq[3];
        w1d_split3(win, bq);
        acc[k] = w1d_mma(a[2], bq[0], acc[k]);          // smallest terms first
        acc[k] = w1d_mma(a[0], bq[2], acc[k]);
        acc[k] = w1d_mma(a[1], bq[1], acc[k]);
        acc[k] = w1d_mma(a[1], bq[0], acc[k]);
        acc[k] = w1d_mma(a[0], bq[1], acc[k]);
        acc[k] = w1d_mma(a[0], bq[0], acc[k]);
      }
    }
  }
  float* rec = partial + (size_t)ch * ((size_t)Cout * Cin * 3 + Cout);
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int o = o0 + (i & 3) + 8 * (i >> 2) + 4 * hh, c = c0 + r;
      if (c < Cin) rec[((size_t)o * Cin + c) * 3 + k] = acc[k][i];
    }
  ab += __shfl_xor(ab, 32, 64);
  if (blockIdx.y == 0 && hh == 0) rec[(size_t)Cout * Cin * 3 + o0 + r] = ab;
}

// data-gradient weights of Conv1d: W'[c][o][k'] = W[o][c][2-k']  (a Conv1d with Cin' = Cout, Cout' = Cin)
__global__ void conv1d_dgrad_pack_kernel(const float* __restrict__ w, float* __restrict__ wt, float* __restrict__ zero_bias,
                                         int cin, int cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cin) zero_bias[i] = 0.f;
  if (i >= cin * cout * 3) return;
  const int k = i % 3, o = (i / 3) % cout, c = i / (3 * cout);
  wt[i] = w[((size_t)o * cin + c) * 3 + (2 - k)];
}

// whole partial record in one launch: elements [0, n0) -> out0, [n0, n0 + n1) -> out1; 64 elements x 4 record groups per
// block, fp64 sums combined in a fixed order (as reduce_wgrad_record_kernel of train_elem.hip)
__global__ __launch_bounds__(256) void reduce_record2_kernel(const float* __restrict__ partial, int nparts, int stride, int n0,
                                                             float* __restrict__ out0, int n1, float* __restrict__ out1) {
  __shared__ double red[4][64];
  const int lane = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + lane, total = n0 + n1;
  double s = 0.0;
  if (e < total) {
    const int k0 = (int)((long)pg * nparts / 4), k1 = (int)((long)(pg + 1) * nparts / 4);
    const float* p = partial + (size_t)k0 * stride + e;
#pragma unroll 8
    for (int k = k0; k < k1; ++k, p += stride) s += (double)*p;
  }
  red[pg][lane] = s;
  __syncthreads();
  if (pg != 0 || e >= total) return;
  const float v = (float)((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
  if (e < n0) out0[e] = v; else out1[e - n0] = v;
}

// ================================================================================================ launchers
constexpr int kCmChunks = 64;   // batch chunks of the channel-major reductions: 16 left the 32->64 weight gradient at 128 blocks on 256 CUs
int cm_chunks(int B) { return B < kCmChunks ? B : kCmChunks; }

hipError_t launch_cm_stats(const float* z, float* partial, int B, int C, int T, hipStream_t s) {
  const int nch = cm_chunks(B), bchunk = (B + nch - 1) / nch;
  hipLaunchKernelGGL(cm_stats_kernel, dim3(C, nch), dim3(256), 0, s, z, partial, B, C, T, bchunk);
  return hipGetLastError();
}
hipError_t launch_cm_bn_relu_drop(const float* z, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, float* h, int B, int C, int T, const DropCfg& dc, hipStream_t s) {
  const size_t n = (size_t)B * C * T;
  hipLaunchKernelGGL(cm_bn_relu_drop_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, mean, invstd, gamma, beta, h, C, T, n, dc);
  return hipGetLastError();
}
hipError_t launch_cm_bn_relu_meant(const float* z, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, float* pooled, int B, int C, int T, hipStream_t s) {
  const int rows = B * C;
  hipLaunchKernelGGL(cm_bn_relu_meant_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, z, mean, invstd, gamma, beta, pooled, C, T, rows);
  return hipGetLastError();
}
hipError_t launch_cm_bn_bwd(int src, const float* z, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, const float* up, float* partial, float* sums, float* dz, int B, int C,
                            int T, const DropCfg& dc, hipStream_t s, const BnSync* sync) {
  const int nch = cm_chunks(B), bchunk = (B + nch - 1) / nch;
  const size_t n = (size_t)B * C * T;
  float inv_n = (float)(1.0 / ((double)B * T));
  if (src == 0)
    hipLaunchKernelGGL(cm_bn_bwd_reduce_kernel<0>, dim3(C, nch), dim3(256), 0, s, z, mean, invstd, gamma, beta, up, partial, B, C, T, bchunk, dc);
  else
    hipLaunchKernelGGL(cm_bn_bwd_reduce_kernel<1>, dim3(C, nch), dim3(256), 0, s, z, mean, invstd, gamma, beta, up, partial, B, C, T, bchunk, dc);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = launch_reduce_partials(partial, nch, C * 2, 1.0f, sums, s, nullptr);
  if (e != hipSuccess) return e;
  const float* sums_a;
  float isc;
  e = bn_sync_sums(sync, sums, C * 2, s, &sums_a, &isc);          // synchronised BatchNorm: global sums, global count
  if (e != hipSuccess) return e;
  inv_n *= isc;
  if (src == 0)
    hipLaunchKernelGGL(cm_bn_bwd_apply_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, mean, invstd, gamma, beta, sums_a, up, dz, C, T, n, inv_n, dc);
  else
    hipLaunchKernelGGL(cm_bn_bwd_apply_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, mean, invstd, gamma, beta, sums_a, up, dz, C, T, n, inv_n, dc);
  return hipGetLastError();
}
// partial: conv1d_wgrad_chunks(B) * (Cout*Cin*3 + Cout) floats
int conv1d_wgrad_chunks(int B) { return B < 256 ? B : 256; }
hipError_t launch_conv1d_wgrad(const float* dz, const float* h, int64_t hsb, int64_t hsc, int64_t hst, float* partial,
                               float* dw, float* db, int B, int Cin, int Cout, int T, hipStream_t s, const AugCfg* aug, int x3) {
  const int nch = conv1d_wgrad_chunks(B), bchunk = (B + nch - 1) / nch;
  if (x3 && Cout % 32 == 0) {      // three-term bf16 matrix-core form (fp32-grade sums)
    const dim3 grid(Cout / 32, (Cin + 31) / 32, nch);
    if (aug && aug->on)
      hipLaunchKernelGGL(conv1d_wgrad_x3_kernel<true>, grid, dim3(64), 0, s, dz, h, hsb, hsc, hst, partial, B, Cin, Cout, T, bchunk, *aug);
    else
      hipLaunchKernelGGL(conv1d_wgrad_x3_kernel<false>, grid, dim3(64), 0, s, dz, h, hsb, hsc, hst, partial, B, Cin, Cout, T, bchunk, AugCfg{});
  } else if (aug && aug->on) {
    if (Cout % 32 != 0) return hipErrorInvalidValue;   // the folded form exists for the MFMA kernel only (layer 1: Cout = 32)
    hipLaunchKernelGGL(conv1d_wgrad_mfma_kernel<true>, dim3(Cout / 32, (Cin + 31) / 32, nch), dim3(64), 0, s, dz, h, hsb, hsc, hst, partial, B, Cin, Cout, T, bchunk, *aug);
  } else if (Cout % 32 == 0)
    hipLaunchKernelGGL(conv1d_wgrad_mfma_kernel<false>, dim3(Cout / 32, (Cin + 31) / 32, nch), dim3(64), 0, s, dz, h, hsb, hsc, hst, partial, B, Cin, Cout, T, bchunk, AugCfg{});
  else
    hipLaunchKernelGGL(conv1d_wgrad_kernel, dim3(Cout / 16, (Cin + 15) / 16, nch), dim3(256), 0, s, dz, h, hsb, hsc, hst, partial, B, Cin, Cout, T, bchunk);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int n = Cout * Cin * 3;
  hipLaunchKernelGGL(reduce_record2_kernel, dim3((n + Cout + 63) / 64), dim3(256), 0, s, partial, nch, n + Cout, n, dw, Cout, db);
  return hipGetLastError();
}
hipError_t launch_conv1d_dgrad_pack(const float* w, float* wt, float* zero_bias, int cin, int cout, hipStream_t s) {
  const int n = cin * cout * 3;
  hipLaunchKernelGGL(conv1d_dgrad_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, wt, zero_bias, cin, cout);
  return hipGetLastError();
}

}  // namespace dfa
