// conv1.hip -- first block of the 2-D CNNs: Conv2d(1->32, 3x3, pad 1) + BN(eval, folded) + ReLU + AvgPool2d(2,1)
// (src/model.py:15-18; the CAE's first block, src/model_cae.py:34-37, uses the (2,2) pool variant).
//
// One input channel and K = 9: no matrix-core shape here; the kernel is bound by writing the 32-channel output
// (1.84 MB/utt in bf16) -- the x tile is staged through LDS so the strided [B,T,F] *view* of the stored [B,F,T]
// tensor (src/predict.py:105) is read along its contiguous axis, and every lane writes its pixel's 32 output
// channels as 16-byte stores into the channels-last activation the MFMA blocks consume.
#include "dfa_internal.h"
#include "rng.h"

namespace dfa {

constexpr int C1_TI = 16;  // pooled rows per tile
constexpr int C1_TF = 16;  // feature columns per tile
constexpr int C1_XR = 2 * C1_TI + 2;
constexpr int C1_XC = C1_TF + 2;

template <typename TX>
__device__ __forceinline__ float load_x(const TX* p);
template <>
__device__ __forceinline__ float load_x<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float load_x<bf16_t>(const bf16_t* p) { return bf16_to_float(*p); }

template <typename TX, typename TO>
__global__ __launch_bounds__(256) void conv1_bn_relu_poolh2_kernel(const TX* __restrict__ x, int64_t sb, int64_t st,
                                                                   int64_t sf, const float* __restrict__ w1,
                                                                   const float* __restrict__ b1, TO* __restrict__ out,
                                                                   int T, int F, int Ho, DropCfg dc) {
  __shared__ float xs[C1_XR][C1_XC + 1];
  const int tid = threadIdx.x;
  const int b = blockIdx.z;
  const int i0 = blockIdx.y * C1_TI, f0 = blockIdx.x * C1_TF;
  const TX* xb = x + (int64_t)b * sb;
  const int t_base = 2 * i0 - 1, f_base = f0 - 1;
  // stage the x tile; walk the contiguous axis with consecutive threads
  const bool t_fast = (st == 1);
  for (int e = tid; e < C1_XR * C1_XC; e += 256) {
    int rr, cc;
    if (t_fast) { cc = e / C1_XR; rr = e - cc * C1_XR; } else { rr = e / C1_XC; cc = e - rr * C1_XC; }
    const int t = t_base + rr, f = f_base + cc;
    float v = 0.f;
    if (t >= 0 && t < T && f >= 0 && f < F) v = load_x<TX>(xb + (int64_t)t * st + (int64_t)f * sf);
    xs[rr][cc] = v;
  }
  __syncthreads();
  const int fi = tid & (C1_TF - 1), ri = tid / C1_TF;
  const int i = i0 + ri, f = f0 + fi;
  if (i >= Ho || f >= F) return;
  float xv[4][3];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int d = 0; d < 3; ++d) xv[a][d] = xs[2 * ri + a][fi + d];

  TO* op = out + (((size_t)b * Ho + i) * F + f) * 32;
  constexpr int VEC = 16 / (int)sizeof(TO);  // output channels per 16-byte store
  float ds[32];                                // train mode: dropout keep-scale of this pixel's 32 channels
  if (dc.thresh != 0) {
#pragma unroll
    for (int c0 = 0; c0 < 32; c0 += 8) drop_scale8(dc, (((uint64_t)b * Ho + i) * F + f) * 32 + c0, ds + c0);
  }
#pragma unroll
  for (int c0 = 0; c0 < 32; c0 += VEC) {
    TO ov[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      const float* wc = w1 + (c0 + c) * 9;  // uniform address -> scalar loads
      float v0 = b1[c0 + c], v1 = v0;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          v0 = fmaf(wc[dy * 3 + d], xv[dy][d], v0);
          v1 = fmaf(wc[dy * 3 + d], xv[dy + 1][d], v1);
        }
      float o = 0.5f * (fmaxf(v0, 0.f) + fmaxf(v1, 0.f));
      if (dc.thresh != 0) o *= ds[c0 + c];
      ov[c] = cvt_out<TO>(o);
    }
    *reinterpret_cast<uint4*>(op + c0) = *reinterpret_cast<const uint4*>(ov);
  }
}

hipError_t launch_conv1(const void* x, int x_dtype, int64_t sb, int64_t st, int64_t sf, const float* w1,
                        const float* b1, void* out, int out_prec, int B, int T, int F, hipStream_t s,
                        const DropCfg* drop) {
  DropCfg dc{};
  if (drop) dc = *drop;
  const int Ho = T / 2;
  dim3 grid((F + C1_TF - 1) / C1_TF, (Ho + C1_TI - 1) / C1_TI, B), block(256);
  if (x_dtype == DFA_DTYPE_F32 && out_prec == DFA_PREC_F32)
    hipLaunchKernelGGL((conv1_bn_relu_poolh2_kernel<float, float>), grid, block, 0, s, (const float*)x, sb, st, sf, w1,
                       b1, (float*)out, T, F, Ho, dc);
  else if (x_dtype == DFA_DTYPE_F32 && out_prec == DFA_PREC_BF16)
    hipLaunchKernelGGL((conv1_bn_relu_poolh2_kernel<float, bf16_t>), grid, block, 0, s, (const float*)x, sb, st, sf,
                       w1, b1, (bf16_t*)out, T, F, Ho, dc);
  else if (x_dtype == DFA_DTYPE_BF16 && out_prec == DFA_PREC_F32)
    hipLaunchKernelGGL((conv1_bn_relu_poolh2_kernel<bf16_t, float>), grid, block, 0, s, (const bf16_t*)x, sb, st, sf,
                       w1, b1, (float*)out, T, F, Ho, dc);
  else
    hipLaunchKernelGGL((conv1_bn_relu_poolh2_kernel<bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)x, sb, st, sf,
                       w1, b1, (bf16_t*)out, T, F, Ho, dc);
  return hipGetLastError();
}

}  // namespace dfa
