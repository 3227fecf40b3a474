// linear.hip -- classifier head: logits[b] = bias + sum_j emb[b][j] * w[j]   (nn.Linear(K, 1), src/model.py:31,39;
// src/model_cnn1d.py:35,45).  One workgroup per utterance; 16-byte loads, wave-64 shuffle reduction, LDS combine.
#include "dfa_internal.h"

namespace dfa {

__global__ __launch_bounds__(256) void linear1_kernel(const float* __restrict__ emb, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ logits,
                                                      int K) {
  __shared__ float part[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* e = emb + (size_t)b * K;
  float acc = 0.f;
  const int K4 = ((((uintptr_t)e | (uintptr_t)w) & 15) == 0) ? (K >> 2) : 0;
  for (int j = tid; j < K4; j += 256) {
    const float4 ev = reinterpret_cast<const float4*>(e)[j], wv = reinterpret_cast<const float4*>(w)[j];
    acc = fmaf(ev.x, wv.x, acc);
    acc = fmaf(ev.y, wv.y, acc);
    acc = fmaf(ev.z, wv.z, acc);
    acc = fmaf(ev.w, wv.w, acc);
  }
  for (int j = 4 * K4 + tid; j < K; j += 256) acc = fmaf(e[j], w[j], acc);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((tid & 63) == 0) part[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0) logits[b] = ((part[0] + part[1]) + (part[2] + part[3])) + bias[0];
}

hipError_t launch_linear(const float* emb, const float* w, const float* bias, float* logits, int B, int K,
                         hipStream_t s) {
  hipLaunchKernelGGL(linear1_kernel, dim3(B), dim3(256), 0, s, emb, w, bias, logits, K);
  return hipGetLastError();
}

// Small batches with the time axis split over workgroups: parts holds nparts slabs of n floats (the canonical chunk sums of the
// time mean, unscaled, stride floats apart).  out[i] = (((0 + p0[i]) + p1[i]) + ...) * inv_h -- the operation sequence of
// the unsplit kernel (running total in chunk order, then the scale), so the embedding is bit-identical to it.
__global__ __launch_bounds__(256) void emb_reduce_kernel(const float* __restrict__ parts, int nparts, size_t stride, size_t n,
                                                         float inv_h, float* __restrict__ out) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < nparts; ++k) {
      const float4 p = *reinterpret_cast<const float4*>(parts + (size_t)k * stride + i);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    *reinterpret_cast<float4*>(out + i) = make_float4(v.x * inv_h, v.y * inv_h, v.z * inv_h, v.w * inv_h);
  } else {
    for (size_t j = i; j < n; ++j) {
      float v = 0.f;
      for (int k = 0; k < nparts; ++k) v += parts[(size_t)k * stride + j];
      out[j] = v * inv_h;
    }
  }
}

hipError_t launch_emb_reduce(const float* parts, int nparts, size_t stride, size_t n, float inv_h, float* out, hipStream_t s) {
  const size_t nthreads = (n + 3) / 4;
  hipLaunchKernelGGL(emb_reduce_kernel, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, s, parts, nparts, stride, n, inv_h, out);
  return hipGetLastError();
}

}  // namespace dfa
