// cae_train.hip -- pieces of the ConvAutoencoder training step (src/train_cae.py:58-82 over src/model_cae.py:32-125)
// that are specific to the decoder: ConvTranspose2d(k2,s2) backward as dense GEMMs on the patch-major view of the
// gradient (gemm_f32.hip), and the 32 -> 1 channel last layer.
#include "dfa_internal.h"

namespace dfa {

template <typename T>
__device__ __forceinline__ void cp8(const T* src, T* dst);
template <>
__device__ __forceinline__ void cp8<float>(const float* src, float* dst) {
  reinterpret_cast<float4*>(dst)[0] = reinterpret_cast<const float4*>(src)[0];
  reinterpret_cast<float4*>(dst)[1] = reinterpret_cast<const float4*>(src)[1];
}
template <>
__device__ __forceinline__ void cp8<bf16_t>(const bf16_t* src, bf16_t* dst) {
  *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
}

// dz[B][2H][Wo][C] (Wo >= 2W; a trailing output_padding column is skipped) -> zp[(b,i,j)][q = 2a+c][C]
template <typename T>
__global__ void pixel_unshuffle_kernel(const T* __restrict__ dz, T* __restrict__ zp, int B, int H, int W, int Wo, int C) {
  const int CG = C >> 3;
  const size_t total = (size_t)B * H * W * 4 * CG;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cg = (int)(i % CG);
  const int q = (int)((i / CG) & 3);
  const size_t p = i / (4 * CG);
  const int j = (int)(p % W);
  const size_t bi = p / W;
  const int ii = (int)(bi % H), b = (int)(bi / H);
  const size_t src = (((size_t)b * 2 * H + 2 * ii + (q >> 1)) * Wo + 2 * j + (q & 1)) * C + cg * 8;
  cp8<T>(dz + src, zp + (p * 4 + q) * C + cg * 8);
}

// torch ConvTranspose2d weight [Cin][Cout][2][2] <-> GEMM operand Wq[Cin][q*Cout + co]
__global__ void convt_w_to_q_kernel(const float* __restrict__ w, float* __restrict__ wq, int cin, int cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cin * cout * 4) return;
  const int q = i & 3, co = (i >> 2) % cout, ci = i / (4 * cout);
  wq[(size_t)ci * 4 * cout + q * cout + co] = w[i];
}
__global__ void convt_q_to_w_kernel(const float* __restrict__ dwq, float* __restrict__ dw, int cin, int cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cin * cout * 4) return;
  const int q = i & 3, co = (i >> 2) % cout, ci = i / (4 * cout);
  dw[i] = dwq[(size_t)ci * 4 * cout + q * cout + co];
}

template <typename T>
__device__ __forceinline__ void ld32(const T* p, float* v);
template <>
__device__ __forceinline__ void ld32<float>(const float* p, float* v) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float4 q = reinterpret_cast<const float4*>(p)[k];
    v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
  }
}
template <>
__device__ __forceinline__ void ld32<bf16_t>(const bf16_t* p, float* v) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint4 q = reinterpret_cast<const uint4*>(p)[k];
    const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[8 * k + 2 * e] = __uint_as_float(u[e] << 16); v[8 * k + 2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
  }
}

// decoder block 4 backward (ConvTranspose2d 32 -> 1): for every d3 pixel (i,j) with its 2x2 patch of drecon
//   dd3[ci] = sum_q g_q * w4[ci][q];   dW4[ci][q] += d3[ci] * g_q;   db4 += sum_q g_q
// Each thread walks many pixels keeping the 129 sums in registers; partial[block][132].
template <typename T>
__global__ __launch_bounds__(256) void cae_dec4_bwd_kernel(const T* __restrict__ d3, const float* __restrict__ w4,
                                                           const float* __restrict__ drecon, T* __restrict__ dd3,
                                                           float* __restrict__ partial, int B, int H3, int W3, int Tt,
                                                           int F) {
  __shared__ float red[4][132];
  const int tid = threadIdx.x;
  float wv[32][4];
#pragma unroll
  for (int ci = 0; ci < 32; ++ci)
#pragma unroll
    for (int q = 0; q < 4; ++q) wv[ci][q] = w4[ci * 4 + q];
  float acc[129];
#pragma unroll
  for (int k = 0; k < 129; ++k) acc[k] = 0.f;
  const size_t npix = (size_t)B * H3 * W3;
  for (size_t p = (size_t)blockIdx.x * 256 + tid; p < npix; p += (size_t)gridDim.x * 256) {
    const int j = (int)(p % W3);
    const size_t bi = p / W3;
    const int i = (int)(bi % H3), b = (int)(bi / H3);
    float g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = 2 * i + (q >> 1), f = 2 * j + (q & 1);
      g[q] = (t < Tt && f < F) ? drecon[((size_t)b * Tt + t) * F + f] : 0.f;
    }
    float v[32], o[32];
    ld32<T>(d3 + p * 32, v);
#pragma unroll
    for (int ci = 0; ci < 32; ++ci) {
      o[ci] = (g[0] * wv[ci][0] + g[1] * wv[ci][1]) + (g[2] * wv[ci][2] + g[3] * wv[ci][3]);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[ci * 4 + q] = fmaf(v[ci], g[q], acc[ci * 4 + q]);
    }
    acc[128] += (g[0] + g[1]) + (g[2] + g[3]);
    T ov[32];
#pragma unroll
    for (int ci = 0; ci < 32; ++ci) ov[ci] = cvt_out<T>(o[ci]);
#pragma unroll
    for (int k = 0; k < (int)(32 * sizeof(T) / 16); ++k)
      reinterpret_cast<uint4*>(dd3 + p * 32)[k] = reinterpret_cast<const uint4*>(ov)[k];
  }
  const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
  for (int k = 0; k < 129; ++k) {
    float s = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) red[wave][k] = s;
  }
  __syncthreads();
  if (tid < 129) partial[(size_t)blockIdx.x * 132 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// fp32 [n] -> T [n]
template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = cvt_out<T>(src[i]);
}

hipError_t launch_pixel_unshuffle(int prec, const void* dz, void* zp, int B, int H, int W, int Wo, int C, hipStream_t s) {
  const size_t total = (size_t)B * H * W * 4 * (C / 8);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (prec == DFA_PREC_BF16)
    hipLaunchKernelGGL(pixel_unshuffle_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)dz, (bf16_t*)zp, B, H, W, Wo, C);
  else
    hipLaunchKernelGGL(pixel_unshuffle_kernel<float>, grid, block, 0, s, (const float*)dz, (float*)zp, B, H, W, Wo, C);
  return hipGetLastError();
}
hipError_t launch_convt_w_to_q(const float* w, float* wq, int cin, int cout, hipStream_t s) {
  const int n = cin * cout * 4;
  hipLaunchKernelGGL(convt_w_to_q_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, wq, cin, cout);
  return hipGetLastError();
}
hipError_t launch_convt_q_to_w(const float* dwq, float* dw, int cin, int cout, hipStream_t s) {
  const int n = cin * cout * 4;
  hipLaunchKernelGGL(convt_q_to_w_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dwq, dw, cin, cout);
  return hipGetLastError();
}
constexpr int kDec4Blocks = 512;
int cae_dec4_bwd_blocks() { return kDec4Blocks; }
hipError_t launch_cae_dec4_bwd(int prec, const void* d3, const float* w4, const float* drecon, void* dd3, float* partial,
                               int B, int H3, int W3, int T, int F, hipStream_t s) {
  if (prec == DFA_PREC_BF16)
    hipLaunchKernelGGL(cae_dec4_bwd_kernel<bf16_t>, dim3(kDec4Blocks), dim3(256), 0, s, (const bf16_t*)d3, w4, drecon, (bf16_t*)dd3, partial, B, H3, W3, T, F);
  else
    hipLaunchKernelGGL(cae_dec4_bwd_kernel<float>, dim3(kDec4Blocks), dim3(256), 0, s, (const float*)d3, w4, drecon, (float*)dd3, partial, B, H3, W3, T, F);
  return hipGetLastError();
}
hipError_t launch_cast_from_f32(int prec, const float* src, void* dst, size_t n, hipStream_t s) {
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (prec == DFA_PREC_BF16) hipLaunchKernelGGL(cast_from_f32_kernel<bf16_t>, grid, block, 0, s, src, (bf16_t*)dst, n);
  else hipLaunchKernelGGL(cast_from_f32_kernel<float>, grid, block, 0, s, src, (float*)dst, n);
  return hipGetLastError();
}

}  // namespace dfa
