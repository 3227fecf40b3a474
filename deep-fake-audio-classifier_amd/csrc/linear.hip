// linear.hip -- classifier head: logits[b] = bias + sum_j emb[b][j] * w[j]   (nn.Linear(K, 1), src/model.py:31,39;
// src/model_cnn1d.py:35,45).  One workgroup per utterance; 16-byte loads, wave-64 shuffle reduction, LDS combine.
#include "dfa_internal.h"

namespace dfa {

// nseg > 1 (small batches, time axis split over workgroups): emb holds the nseg canonical chunk sums of the time mean per
// element, seg_stride floats apart, unscaled; they are added in chunk order and scaled by inv_h -- the operation sequence of
// the unsplit kernel -- and, when emb_out != nullptr, the means are written there: the embedding the caller asked for.
__global__ __launch_bounds__(256) void linear1_kernel(const float* __restrict__ emb, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ logits,
                                                      int K, int nseg, size_t seg_stride, float* __restrict__ emb_out,
                                                      float inv_h) {
  __shared__ float part[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* e = emb + (size_t)b * K;
  float acc = 0.f;
  if (nseg > 1) {
    for (int j = tid; j < K; j += 256) {
      float v = 0.f;
      for (int sg = 0; sg < nseg; ++sg) v += e[(size_t)sg * seg_stride + j];
      v *= inv_h;
      if (emb_out) emb_out[(size_t)b * K + j] = v;
      acc = fmaf(v, w[j], acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((tid & 63) == 0) part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) logits[b] = ((part[0] + part[1]) + (part[2] + part[3])) + bias[0];
    return;
  }
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
                         hipStream_t s, int nseg, size_t seg_stride, float* emb_out, float inv_h) {
  hipLaunchKernelGGL(linear1_kernel, dim3(B), dim3(256), 0, s, emb, w, bias, logits, K, nseg, seg_stride, emb_out, inv_h);
  return hipGetLastError();
}

}  // namespace dfa
