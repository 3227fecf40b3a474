// gemm_f32.hip -- small general GEMM on the fp32 matrix cores, used by the ConvTranspose2d backward of the
// auto-encoder (autograd of src/model_cae.py:63-79 inside loss.backward(), src/train_cae.py:71).
//
// kernel == stride makes ConvTranspose2d(k2,s2) a plain matrix product on the "patch-major" view of its output
// (each input pixel owns its 2x2 output patch: Z[p][q*Cout+co]):
//     forward   Z  = X . Wm            X [P x Cin],  Wm [Cin x 4Cout] (torch's weight[ci][co][a][c] IS row-major Wm)
//     dgrad     dX = dZ . Wm^T
//     wgrad     dWm = X^T . dZ         (K = P pixels: split over workgroups, fixed-order reduction of the partials)
// so the backward needs no gather kernel of its own: one pixel-unshuffle copy of dZ, then this GEMM twice.
//
// C[M][N] = sum_k A(m,k) * B(k,n) with A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn] (element strides, so
// transposes are free), inputs fp32 or bf16 (widened while staging), fp32 accumulate on v_mfma_f32_32x32x2_f32.
// Tile 64 x 64 x 32 per 256-thread workgroup (each wave one 32 x 32 accumulator); grid.z splits K and writes
// partial[z][M][N].
#include "dfa_internal.h"

namespace dfa {

template <typename T>
__device__ __forceinline__ float ldg1(const T* p);
template <>
__device__ __forceinline__ float ldg1<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ldg1<bf16_t>(const bf16_t* p) { return bf16_to_float(*p); }

constexpr int GM = 64, GN = 64, GK = 32;

template <typename TA, typename TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const TA* __restrict__ A, int64_t sam, int64_t sak,
                                                       const TB* __restrict__ Bm, int64_t sbk, int64_t sbn,
                                                       float* __restrict__ C, int M, int N, int K, int kchunk) {
  __shared__ float As[GM][GK + 1];
  __shared__ float Bs[GK][GN + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * GM, n0 = blockIdx.x * GN;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int k0 = blockIdx.z * kchunk, k1 = min(K, k0 + kchunk);
  f32x16_t acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // staging order follows whichever stride is 1 so that global reads are contiguous
  const bool a_kfast = (sak == 1), b_nfast = (sbn == 1);
  for (int kk = k0; kk < k1; kk += GK) {
    __syncthreads();
    for (int e = tid; e < GM * GK; e += 256) {
      int mm, k;
      if (a_kfast) { mm = e / GK; k = e - mm * GK; } else { k = e / GM; mm = e - k * GM; }
      const int m = m0 + mm, kg = kk + k;
      As[mm][k] = (m < M && kg < k1) ? ldg1<TA>(A + (int64_t)m * sam + (int64_t)kg * sak) : 0.f;
    }
    for (int e = tid; e < GK * GN; e += 256) {
      int k, nn;
      if (b_nfast) { k = e / GN; nn = e - k * GN; } else { nn = e / GK; k = e - nn * GK; }
      const int n = n0 + nn, kg = kk + k;
      Bs[k][nn] = (n < N && kg < k1) ? ldg1<TB>(Bm + (int64_t)kg * sbk + (int64_t)n * sbn) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GK; k += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[wm + r][k + h], Bs[k + h][wn + r], acc, 0, 0, 0);
  }
  float* Cz = C + (size_t)blockIdx.z * M * N;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = m0 + wm + (i & 3) + 8 * (i >> 2) + 4 * h, n = n0 + wn + r;
    if (m < M && n < N) Cz[(size_t)m * N + n] = acc[i];
  }
}

// C (or, when ksplit > 1, partial[ksplit][M][N] which the caller reduces) = A . B
hipError_t launch_gemm_f32(int a_bf16, const void* A, int64_t sam, int64_t sak, int b_bf16, const void* Bm, int64_t sbk,
                           int64_t sbn, float* C, int M, int N, int K, int ksplit, hipStream_t s) {
  const int kchunk = ((K + ksplit - 1) / ksplit + GK - 1) / GK * GK;
  dim3 grid((N + GN - 1) / GN, (M + GM - 1) / GM, ksplit), block(256);
  if (!a_bf16 && !b_bf16)
    hipLaunchKernelGGL((gemm_f32_kernel<float, float>), grid, block, 0, s, (const float*)A, sam, sak, (const float*)Bm, sbk, sbn, C, M, N, K, kchunk);
  else if (a_bf16 && !b_bf16)
    hipLaunchKernelGGL((gemm_f32_kernel<bf16_t, float>), grid, block, 0, s, (const bf16_t*)A, sam, sak, (const float*)Bm, sbk, sbn, C, M, N, K, kchunk);
  else if (!a_bf16 && b_bf16)
    hipLaunchKernelGGL((gemm_f32_kernel<float, bf16_t>), grid, block, 0, s, (const float*)A, sam, sak, (const bf16_t*)Bm, sbk, sbn, C, M, N, K, kchunk);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)A, sam, sak, (const bf16_t*)Bm, sbk, sbn, C, M, N, K, kchunk);
  return hipGetLastError();
}

}  // namespace dfa
