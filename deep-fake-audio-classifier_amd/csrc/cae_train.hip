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
// MSE form (drecon == nullptr; src/train_cae.py:67-68 loss = MSELoss(recon, x)): the upstream gradient never exists in memory --
// g_q = mse_scale * (recon_q - x_q) with recon_q = b4 + sum_ci d3[ci] * w4[ci][q] recomputed from the d3 pixel the thread holds
// anyway (128 FMAs) and x read through the caller's strides; mse_scale = 2 / (B*T*F).
struct MseSrc {
  const void* x;
  int x_bf16;
  int64_t sb, st, sf;
  float b4, scale;
  const float* b4_dev;     // device pointer to decoder.9.bias (read once per thread)
};
// MODE 0: upstream gradient from drecon; 1: MSE form, x bf16; 2: MSE form, x fp32 (template: no per-element branches, so the next
// pixel's loads -- d3 row and, in the MSE form, its four x values -- can be requested one pixel AHEAD with clamped addresses)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void cae_dec4_bwd_kernel(const T* __restrict__ d3, const float* __restrict__ w4,
                                                           const float* __restrict__ drecon, T* __restrict__ dd3,
                                                           float* __restrict__ partial, int B, int H3, int W3, int Tt,
                                                           int F, MseSrc mse) {
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
  const float bias4 = MODE == 0 ? 0.f : mse.b4_dev[0];
  const unsigned npix = (unsigned)((size_t)B * H3 * W3);                 // < 2^31 (launcher-checked): 32-bit pixel arithmetic
  const unsigned stride = gridDim.x * 256u;
  constexpr int NRAW = 32 * (int)sizeof(T) / 16;
  uint4 rawc[NRAW], rawn[NRAW];
  float gc[4], gn[4];                                                    // MODE 0: drecon values; MODE 1/2: raw x values (0 outside the image)
  unsigned okc = 0, okn = 0;                                             // bit q: patch element q lies inside [T] x [F]
  auto fetch = [&](unsigned p, uint4 (&raw)[NRAW], float (&g)[4], unsigned& ok) {
    const unsigned pc = min(p, npix - 1);                                // clamped: a prefetch past the end re-reads the last pixel
#pragma unroll
    for (int k = 0; k < NRAW; ++k) raw[k] = reinterpret_cast<const uint4*>(d3 + (size_t)pc * 32)[k];
    const unsigned j = pc % (unsigned)W3, bi = pc / (unsigned)W3;
    const unsigned i = bi % (unsigned)H3, b = bi / (unsigned)H3;
    ok = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int t = 2 * (int)i + (q >> 1), f = 2 * (int)j + (q & 1);
      const bool in = t < Tt && f < F;
      ok |= in ? (1u << q) : 0u;
      const int tc = min(t, Tt - 1), fc = min(f, F - 1);
      if constexpr (MODE == 0) g[q] = drecon[((size_t)b * Tt + tc) * F + fc];
      else {
        const int64_t off = (int64_t)b * mse.sb + (int64_t)tc * mse.st + (int64_t)fc * mse.sf;
        if constexpr (MODE == 1) g[q] = bf16_to_float(((const bf16_t*)mse.x)[off]);
        else g[q] = ((const float*)mse.x)[off];
      }
    }
  };
  unsigned p = blockIdx.x * 256u + tid;
  if (p < npix) fetch(p, rawc, gc, okc);
  for (; p < npix; p += stride) {
    fetch(p + stride, rawn, gn, okn);                                    // next pixel: in flight under this pixel's ~400 FMAs
    float v[32], o[32], g[4];
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int k = 0; k < NRAW; ++k) {
        const unsigned u[4] = {rawc[k].x, rawc[k].y, rawc[k].z, rawc[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[8 * k + 2 * e] = __uint_as_float(u[e] << 16); v[8 * k + 2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
      }
    } else {
#pragma unroll
      for (int k = 0; k < NRAW; ++k) {
        v[4 * k] = __uint_as_float(rawc[k].x); v[4 * k + 1] = __uint_as_float(rawc[k].y);
        v[4 * k + 2] = __uint_as_float(rawc[k].z); v[4 * k + 3] = __uint_as_float(rawc[k].w);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool in = (okc >> q) & 1u;
      if constexpr (MODE == 0) g[q] = in ? gc[q] : 0.f;
      else {
        float r = bias4;
#pragma unroll
        for (int ci = 0; ci < 32; ++ci) r = fmaf(v[ci], wv[ci][q], r);
        g[q] = in ? mse.scale * (r - gc[q]) : 0.f;
      }
    }
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
    for (int k = 0; k < NRAW; ++k)
      reinterpret_cast<uint4*>(dd3 + (size_t)p * 32)[k] = reinterpret_cast<const uint4*>(ov)[k];
#pragma unroll
    for (int k = 0; k < NRAW; ++k) rawc[k] = rawn[k];
#pragma unroll
    for (int q = 0; q < 4; ++q) gc[q] = gn[q];
    okc = okn;
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
                               int B, int H3, int W3, int T, int F, hipStream_t s, const MseArgs* mse_args) {
  MseSrc mse{};
  if (!drecon) {
    if (!mse_args) return hipErrorInvalidValue;
    mse.x = mse_args->x; mse.x_bf16 = mse_args->x_bf16; mse.sb = mse_args->sb; mse.st = mse_args->st; mse.sf = mse_args->sf;
    mse.b4 = 0.f; mse.scale = 2.0f / ((float)B * (float)T * (float)F);
    mse.b4_dev = mse_args->b4_dev;
  }
  if ((size_t)B * H3 * W3 >= ((size_t)1 << 31)) return hipErrorInvalidValue;      // 32-bit pixel indices
  const int mode = drecon ? 0 : (mse.x_bf16 ? 1 : 2);
#define DFA_D4B(TT, MODE) hipLaunchKernelGGL((cae_dec4_bwd_kernel<TT, MODE>), dim3(kDec4Blocks), dim3(256), 0, s, (const TT*)d3, w4, drecon, (TT*)dd3, partial, B, H3, W3, T, F, mse)
  if (prec == DFA_PREC_BF16) { if (mode == 0) DFA_D4B(bf16_t, 0); else if (mode == 1) DFA_D4B(bf16_t, 1); else DFA_D4B(bf16_t, 2); }
  else { if (mode == 0) DFA_D4B(float, 0); else if (mode == 1) DFA_D4B(float, 1); else DFA_D4B(float, 2); }
#undef DFA_D4B
  return hipGetLastError();
}
// MSELoss(recon, x) forward + backward in one pass (src/train_cae.py:67-68, criterion at :203): every block sums its squared
// differences in a fixed order and writes drecon = 2 (recon - x) / n; a second one-block launch adds the block sums in index order.
__global__ __launch_bounds__(256) void mse_fwd_bwd_kernel(const float* __restrict__ recon, const void* __restrict__ x, int x_bf16,
                                                          int64_t sb, int64_t st, int64_t sf, int T, int F, size_t n, float scale,
                                                          float* __restrict__ partial, float* __restrict__ drecon) {
  __shared__ float red[4];
  const int tid = threadIdx.x;
  float acc = 0.f;
  const size_t per = (n + gridDim.x - 1) / gridDim.x;
  const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  for (size_t i = lo + tid; i < hi; i += 256) {
    const int f = (int)(i % F);
    const size_t bt = i / F;
    const int t = (int)(bt % T);
    const int64_t b = (int64_t)(bt / T);
    const int64_t off = b * sb + (int64_t)t * st + (int64_t)f * sf;
    const float xv = x_bf16 ? bf16_to_float(((const bf16_t*)x)[off]) : ((const float*)x)[off];
    const float d = recon[i] - xv;
    acc = fmaf(d, d, acc);
    if (drecon) drecon[i] = scale * d;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void mse_finish_kernel(const float* __restrict__ partial, int nblk, double inv_n, float* __restrict__ loss) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < nblk; i += 256) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] * inv_n);
}
hipError_t launch_mse_fwd_bwd(const float* recon, const void* x, int x_bf16, int64_t sb, int64_t st, int64_t sf, int B, int T, int F,
                              float* partial, float* loss, float* drecon, hipStream_t s) {
  const size_t n = (size_t)B * T * F;
  hipLaunchKernelGGL(mse_fwd_bwd_kernel, dim3(kMseBlocks), dim3(256), 0, s, recon, x, x_bf16, sb, st, sf, T, F, n, (float)(2.0 / (double)n),
                     partial, drecon);
  if (loss) hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, s, partial, kMseBlocks, 1.0 / (double)n, loss);
  return hipGetLastError();
}
hipError_t launch_cast_from_f32(int prec, const float* src, void* dst, size_t n, hipStream_t s) {
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  if (prec == DFA_PREC_BF16) hipLaunchKernelGGL(cast_from_f32_kernel<bf16_t>, grid, block, 0, s, src, (bf16_t*)dst, n);
  else hipLaunchKernelGGL(cast_from_f32_kernel<float>, grid, block, 0, s, src, (float*)dst, n);
  return hipGetLastError();
}

}  // namespace dfa
