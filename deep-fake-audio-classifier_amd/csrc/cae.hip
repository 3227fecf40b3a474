// cae.hip -- the ConvAutoencoder pieces that are not matrix-core shaped (src/model_cae.py:83-125,
// src/evaluation_cae.py:52-53):
//   * encoder block 1: Conv2d(1->32,3x3) + BN + ReLU + AvgPool2d(2)  with the FeatureNormalizer z-score
//     (src/dataset_cae.py:37-41) fused into the load: x_n = (x - mean[f]) / std[f];
//   * the bias-only column that ConvTranspose2d(output_padding=(0,1)) appends (model_cae.py:68-69);
//   * decoder block 4: ConvTranspose2d(32->1, k2 s2), zero padding of the time axis to T (model_cae.py:113-119),
//     and the per-sample MSE against the (normalised) input, reduced in a fixed order (no atomics);
//   * the latent map export to the reference's NCHW float32 layout.
#include "dfa_internal.h"

namespace dfa {

template <typename TX>
__device__ __forceinline__ float ld_x(const TX* p);
template <>
__device__ __forceinline__ float ld_x<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld_x<bf16_t>(const bf16_t* p) { return bf16_to_float(*p); }

constexpr int E1_TI = 16, E1_TJ = 16, E1_XR = 2 * E1_TI + 2, E1_XC = 2 * E1_TJ + 2;

template <typename TX, typename TO>
__global__ __launch_bounds__(256) void cae_enc1_kernel(const TX* __restrict__ x, int64_t sb, int64_t st, int64_t sf,
                                                       const float* __restrict__ mu, const float* __restrict__ sigma,
                                                       const float* __restrict__ w1, const float* __restrict__ b1,
                                                       TO* __restrict__ out, int T, int F, int Ho, int Wo) {
  __shared__ float xs[E1_XR][E1_XC + 1];
  const int tid = threadIdx.x, b = blockIdx.z;
  const int i0 = blockIdx.y * E1_TI, j0 = blockIdx.x * E1_TJ;
  const TX* xb = x + (int64_t)b * sb;
  const int t_base = 2 * i0 - 1, f_base = 2 * j0 - 1;
  const bool t_fast = (st == 1);
  for (int e = tid; e < E1_XR * E1_XC; e += 256) {
    int rr, cc;
    if (t_fast) { cc = e / E1_XR; rr = e - cc * E1_XR; } else { rr = e / E1_XC; cc = e - rr * E1_XC; }
    const int t = t_base + rr, f = f_base + cc;
    float v = 0.f;  // conv zero padding applies to the NORMALISED input
    if (t >= 0 && t < T && f >= 0 && f < F) {
      v = ld_x<TX>(xb + (int64_t)t * st + (int64_t)f * sf);
      if (mu) v = (v - mu[f]) / sigma[f];
    }
    xs[rr][cc] = v;
  }
  __syncthreads();
  const int jj = tid & (E1_TJ - 1), ii = tid / E1_TJ;
  const int i = i0 + ii, j = j0 + jj;
  if (i >= Ho || j >= Wo) return;
  float xv[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int d = 0; d < 4; ++d) xv[a][d] = xs[2 * ii + a][2 * jj + d];
  TO* op = out + (((size_t)b * Ho + i) * Wo + j) * 32;
  constexpr int VEC = 16 / (int)sizeof(TO);
#pragma unroll
  for (int c0 = 0; c0 < 32; c0 += VEC) {
    TO ov[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
      const float* wc = w1 + (c0 + c) * 9;
      const float bb = b1[c0 + c];
      float v00 = bb, v01 = bb, v10 = bb, v11 = bb;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const float wk = wc[dy * 3 + d];
          v00 = fmaf(wk, xv[dy][d], v00);
          v01 = fmaf(wk, xv[dy][d + 1], v01);
          v10 = fmaf(wk, xv[dy + 1][d], v10);
          v11 = fmaf(wk, xv[dy + 1][d + 1], v11);
        }
      ov[c] = cvt_out<TO>(0.25f * ((fmaxf(v00, 0.f) + fmaxf(v01, 0.f)) + (fmaxf(v10, 0.f) + fmaxf(v11, 0.f))));
    }
    *reinterpret_cast<uint4*>(op + c0) = *reinterpret_cast<const uint4*>(ov);
  }
}

template <typename T>
__global__ void cae_opad_col_kernel(T* __restrict__ out, const float* __restrict__ bias, int rows, int Wo, int C,
                                    int no_relu) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // over rows*C
  if (i >= rows * C) return;
  const int row = i / C, c = i - row * C;
  out[((size_t)row * Wo + (Wo - 1)) * C + c] = cvt_out<T>(no_relu ? bias[c] : fmaxf(bias[c], 0.f));
}

template <typename T>
__device__ __forceinline__ void load32(const T* p, float* v);
template <>
__device__ __forceinline__ void load32<float>(const float* p, float* v) {
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float4 q = reinterpret_cast<const float4*>(p)[k];
    v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
  }
}
template <>
__device__ __forceinline__ void load32<bf16_t>(const bf16_t* p, float* v) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint4 q = reinterpret_cast<const uint4*>(p)[k];
    const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[8 * k + 2 * e] = __uint_as_float(u[e] << 16);
      v[8 * k + 2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u);
    }
  }
}

// decoder block 4 + zero time padding + squared error.  One thread per input pixel (i, j) of d3, plus "virtual"
// rows i >= H3 that only cover the zero-padded tail of the reconstruction.
template <typename T, typename TX>
__global__ __launch_bounds__(256) void cae_dec4_mse_kernel(const T* __restrict__ d3, const float* __restrict__ w4,
                                                           const float* __restrict__ b4, const TX* __restrict__ x,
                                                           int64_t sb, int64_t st, int64_t sf,
                                                           const float* __restrict__ mu,
                                                           const float* __restrict__ sigma, float* __restrict__ recon,
                                                           float* __restrict__ partial, int H3, int W3, int Tt, int F) {
  __shared__ float red[4];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int NI = (Tt + 1) / 2;
  const int e = blockIdx.x * 256 + tid;
  float err = 0.f;
  if (e < NI * W3) {
    const int i = e / W3, j = e - i * W3;
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < H3) {
      float v[32];
      load32<T>(d3 + (((size_t)b * H3 + i) * W3 + j) * 32, v);
      const float bb = b4[0];
#pragma unroll
      for (int q = 0; q < 4; ++q) r[q] = bb;
#pragma unroll
      for (int ci = 0; ci < 32; ++ci)
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = fmaf(v[ci], w4[ci * 4 + q], r[q]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int t = 2 * i + a;
      if (t < Tt) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int f = 2 * j + c;
          float xn = ld_x<TX>(x + (int64_t)b * sb + (int64_t)t * st + (int64_t)f * sf);
          if (mu) xn = (xn - mu[f]) / sigma[f];
          const float d = r[2 * a + c] - xn;
          err = fmaf(d, d, err);
        }
        if (recon) *reinterpret_cast<float2*>(recon + ((size_t)b * Tt + t) * F + 2 * j) = make_float2(r[2 * a], r[2 * a + 1]);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) err += __shfl_down(err, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = err;
  __syncthreads();
  if (tid == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void cae_mse_finalize_kernel(const float* __restrict__ partial, int nblk, float inv_n, float* __restrict__ mse,
                                        int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0.0;
  for (int k = 0; k < nblk; ++k) s += (double)partial[(size_t)b * nblk + k];
  mse[b] = (float)(s * (double)inv_n);
}

template <typename T>
__device__ __forceinline__ float to_float(T v);
template <>
__device__ __forceinline__ float to_float<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_float<bf16_t>(bf16_t v) { return bf16_to_float(v); }

template <typename T>
__global__ void cae_latent_export_kernel(const T* __restrict__ lat, float* __restrict__ out, int B, int HW, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // over out elements [B][C][HW]
  if (i >= (size_t)B * C * HW) return;
  const int p = (int)(i % HW);
  const int c = (int)((i / HW) % C);
  const int b = (int)(i / ((size_t)HW * C));
  out[i] = to_float<T>(lat[((size_t)b * HW + p) * C + c]);
}

hipError_t launch_cae_enc1(const void* x, int x_dtype, int64_t sb, int64_t st, int64_t sf, const float* mu,
                           const float* sigma, const float* w1, const float* b1, void* out, int prec, int B, int T, int F,
                           hipStream_t s) {
  const int Ho = T / 2, Wo = F / 2;
  dim3 grid((Wo + E1_TJ - 1) / E1_TJ, (Ho + E1_TI - 1) / E1_TI, B), block(256);
  if (x_dtype == DFA_DTYPE_F32 && prec == DFA_PREC_F32)
    hipLaunchKernelGGL((cae_enc1_kernel<float, float>), grid, block, 0, s, (const float*)x, sb, st, sf, mu, sigma, w1, b1, (float*)out, T, F, Ho, Wo);
  else if (x_dtype == DFA_DTYPE_F32)
    hipLaunchKernelGGL((cae_enc1_kernel<float, bf16_t>), grid, block, 0, s, (const float*)x, sb, st, sf, mu, sigma, w1, b1, (bf16_t*)out, T, F, Ho, Wo);
  else if (prec == DFA_PREC_F32)
    hipLaunchKernelGGL((cae_enc1_kernel<bf16_t, float>), grid, block, 0, s, (const bf16_t*)x, sb, st, sf, mu, sigma, w1, b1, (float*)out, T, F, Ho, Wo);
  else
    hipLaunchKernelGGL((cae_enc1_kernel<bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)x, sb, st, sf, mu, sigma, w1, b1, (bf16_t*)out, T, F, Ho, Wo);
  return hipGetLastError();
}

hipError_t launch_cae_opad_col(void* out, const float* bias, int prec, int rows, int Wo, int C, hipStream_t s,
                               int no_relu) {
  const int n = rows * C;
  if (prec == DFA_PREC_BF16)
    hipLaunchKernelGGL(cae_opad_col_kernel<bf16_t>, dim3((n + 255) / 256), dim3(256), 0, s, (bf16_t*)out, bias, rows, Wo, C, no_relu);
  else
    hipLaunchKernelGGL(cae_opad_col_kernel<float>, dim3((n + 255) / 256), dim3(256), 0, s, (float*)out, bias, rows, Wo, C, no_relu);
  return hipGetLastError();
}

int cae_dec4_blocks(int T, int W3) { return (((T + 1) / 2) * W3 + 255) / 256; }

hipError_t launch_cae_dec4_mse(const void* d3, int prec, const float* w4, const float* b4, const void* x, int x_dtype,
                               int64_t sb, int64_t st, int64_t sf, const float* mu, const float* sigma, float* recon,
                               float* partial, float* mse, int B, int H3, int W3, int T, int F, hipStream_t s) {
  const int nblk = cae_dec4_blocks(T, W3);
  dim3 grid(nblk, B), block(256);
  if (prec == DFA_PREC_F32 && x_dtype == DFA_DTYPE_F32)
    hipLaunchKernelGGL((cae_dec4_mse_kernel<float, float>), grid, block, 0, s, (const float*)d3, w4, b4, (const float*)x, sb, st, sf, mu, sigma, recon, partial, H3, W3, T, F);
  else if (prec == DFA_PREC_F32)
    hipLaunchKernelGGL((cae_dec4_mse_kernel<float, bf16_t>), grid, block, 0, s, (const float*)d3, w4, b4, (const bf16_t*)x, sb, st, sf, mu, sigma, recon, partial, H3, W3, T, F);
  else if (x_dtype == DFA_DTYPE_F32)
    hipLaunchKernelGGL((cae_dec4_mse_kernel<bf16_t, float>), grid, block, 0, s, (const bf16_t*)d3, w4, b4, (const float*)x, sb, st, sf, mu, sigma, recon, partial, H3, W3, T, F);
  else
    hipLaunchKernelGGL((cae_dec4_mse_kernel<bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)d3, w4, b4, (const bf16_t*)x, sb, st, sf, mu, sigma, recon, partial, H3, W3, T, F);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (mse) {
    hipLaunchKernelGGL(cae_mse_finalize_kernel, dim3((B + 255) / 256), dim3(256), 0, s, partial, nblk, 1.0f / ((float)T * (float)F), mse, B);
    e = hipGetLastError();
  }
  return e;
}

hipError_t launch_cae_mse_finalize(const float* partial, int nblk, float inv_n, float* mse, int B, hipStream_t s) {
  hipLaunchKernelGGL(cae_mse_finalize_kernel, dim3((B + 255) / 256), dim3(256), 0, s, partial, nblk, inv_n, mse, B);
  return hipGetLastError();
}

hipError_t launch_cae_latent_export(const void* lat, int prec, float* out, int B, int HW, int C, hipStream_t s) {
  const size_t n = (size_t)B * C * HW;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (prec == DFA_PREC_BF16)
    hipLaunchKernelGGL(cae_latent_export_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)lat, out, B, HW, C);
  else
    hipLaunchKernelGGL(cae_latent_export_kernel<float>, grid, block, 0, s, (const float*)lat, out, B, HW, C);
  return hipGetLastError();
}

}  // namespace dfa
