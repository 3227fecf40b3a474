// conv1d.hip -- CNN1D blocks: Conv1d(k=3, pad 1) + BatchNorm1d(eval, folded) + ReLU, the last one fused with
// AdaptiveAvgPool1d(1) (mean over T)   (src/model_cnn1d.py:17-34,40-43).
//
// 30.8 MFLOP and 231 KB of input per utterance: the path is bound by reading x once.  Activations are
// channel-major [B][C][T] fp32 -- which IS the stored feature layout [B,180,321] (src/dataset.py:52), so layer 1
// reads the caller's tensor in place through its strides (the model's transpose(1,2), model_cnn1d.py:40, is a view).
// Tile = 32 output channels x 64 frames per workgroup; input channels are streamed through LDS 16 at a time
// (x slab [16][66] + weight slab [16][3][32]); every thread owns 2 channels x 4 consecutive frames.
#include "dfa_internal.h"
#include "rng.h"

namespace dfa {

constexpr int C1D_OT = 32, C1D_TT = 64, C1D_CC = 16;

// AUG (train-mode layer 1 only): x is read through the armed train-time augmentation (rng.h aug_apply; channel = feature dim,
// dfa_cnn1d_set_train_augment) -- the element the stand-alone dfa_augment_batch pass would have written, never materialised.
template <bool MEAN, bool RELU = true, bool AUG = false>
__global__ __launch_bounds__(256) void conv1d_k3_bn_relu_kernel(const float* __restrict__ x, int64_t sb, int64_t sc,
                                                                 int64_t st, const float* __restrict__ w,
                                                                 const float* __restrict__ bias,
                                                                 float* __restrict__ out, int Cin, int Cout, int T,
                                                                 float inv_t, AugCfg aug = AugCfg{}) {
  __shared__ float xs[C1D_CC][C1D_TT + 4];
  __shared__ float ws[C1D_CC][3][C1D_OT];
  __shared__ float red[C1D_OT][17];
  const int tid = threadIdx.x;
  const int tq = tid & 15, oq = tid >> 4;       // 16 frame-quads x 16 channel-pairs
  const int b = blockIdx.z, o0 = blockIdx.y * C1D_OT;
  const float* xb = x + (int64_t)b * sb;
  const float b0 = bias[o0 + 2 * oq], b1 = bias[o0 + 2 * oq + 1];
  float msum0 = 0.f, msum1 = 0.f;

  const int tile_lo = MEAN ? 0 : blockIdx.x;
  const int tile_hi = MEAN ? (T + C1D_TT - 1) / C1D_TT : blockIdx.x + 1;
  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const int t0 = tile * C1D_TT;
    float acc[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[0][j] = b0; acc[1][j] = b1; }
    for (int c0 = 0; c0 < Cin; c0 += C1D_CC) {
      __syncthreads();
      for (int e = tid; e < C1D_CC * (C1D_TT + 2); e += 256) {
        const int c = e / (C1D_TT + 2), tt = e - c * (C1D_TT + 2);
        const int t = t0 - 1 + tt, ci = c0 + c;
        if constexpr (AUG)
          xs[c][tt] = (ci < Cin && t >= 0 && t < T) ? aug_apply(aug, xb[(int64_t)ci * sc + (int64_t)aug_src_t(aug, t) * st], b, t, ci) : 0.f;
        else
          xs[c][tt] = (ci < Cin && t >= 0 && t < T) ? xb[(int64_t)ci * sc + (int64_t)t * st] : 0.f;
      }
      for (int e = tid; e < C1D_CC * 3 * C1D_OT; e += 256) {
        const int o = e & (C1D_OT - 1), k = (e / C1D_OT) % 3, c = e / (3 * C1D_OT);
        const int ci = c0 + c;
        ws[c][k][o] = (ci < Cin) ? w[((size_t)(o0 + o) * Cin + ci) * 3 + k] : 0.f;
      }
      __syncthreads();
#pragma unroll 4
      for (int c = 0; c < C1D_CC; ++c) {
        float xv[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) xv[j] = xs[c][4 * tq + j];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const float w0 = ws[c][k][2 * oq], w1 = ws[c][k][2 * oq + 1];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[0][j] = fmaf(w0, xv[j + k], acc[0][j]);
            acc[1][j] = fmaf(w1, xv[j + k], acc[1][j]);
          }
        }
      }
    }
    if (MEAN) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (t0 + 4 * tq + j < T) { msum0 += fmaxf(acc[0][j], 0.f); msum1 += fmaxf(acc[1][j], 0.f); }
    } else {
#pragma unroll
      for (int oo = 0; oo < 2; ++oo) {
        float* orow = out + ((size_t)b * Cout + o0 + 2 * oq + oo) * T;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int t = t0 + 4 * tq + j;
          if (t < T) orow[t] = RELU ? fmaxf(acc[oo][j], 0.f) : acc[oo][j];
        }
      }
    }
  }
  if (MEAN) {  // fixed-order reduction over the 16 frame-quads -> pooled[b][o] (deterministic, no atomics)
    red[2 * oq][tq] = msum0;
    red[2 * oq + 1][tq] = msum1;
    __syncthreads();
    if (tid < C1D_OT) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) s += red[tid][j];
      out[(size_t)b * Cout + o0 + tid] = s * inv_t;
    }
  }
}

// fold BN1d into conv weights (same [Cout][Cin][3] layout) and bias
__global__ void fold_conv1d_kernel(const float* __restrict__ w, const float* __restrict__ b, const float* __restrict__ g,
                                   const float* __restrict__ beta, const float* __restrict__ mean,
                                   const float* __restrict__ var, float* __restrict__ wf, float* __restrict__ bf,
                                   int cin, int cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cout * cin * 3) {
    const int o = i / (cin * 3);
    wf[i] = w[i] * (g[o] / sqrtf(var[o] + kBnEps));
  }
  if (i < cout) bf[i] = (b[i] - mean[i]) * (g[i] / sqrtf(var[i] + kBnEps)) + beta[i];
}

hipError_t launch_fold_conv1d(const float* w, const float* b, const float* g, const float* beta, const float* mean,
                              const float* var, float* wf, float* bf, int cin, int cout, hipStream_t s) {
  const int n = cout * cin * 3;
  hipLaunchKernelGGL(fold_conv1d_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, b, g, beta, mean, var, wf, bf, cin,
                     cout);
  return hipGetLastError();
}

hipError_t launch_conv1d(const float* x, int64_t sb, int64_t sc, int64_t st, const float* w, const float* bias,
                         float* out, int B, int Cin, int Cout, int T, bool mean, hipStream_t s, bool relu, const AugCfg* aug) {
  if (aug && aug->on) {   // train-mode layer 1 with the augmentation folded into the x loads
    if (relu || mean) return hipErrorInvalidValue;
    hipLaunchKernelGGL((conv1d_k3_bn_relu_kernel<false, false, true>), dim3((T + C1D_TT - 1) / C1D_TT, Cout / C1D_OT, B), dim3(256),
                       0, s, x, sb, sc, st, w, bias, out, Cin, Cout, T, 0.f, *aug);
    return hipGetLastError();
  }
  if (!relu) {   // raw convolution (+ bias): train-mode forward and the data-gradient convolution
    hipLaunchKernelGGL((conv1d_k3_bn_relu_kernel<false, false>), dim3((T + C1D_TT - 1) / C1D_TT, Cout / C1D_OT, B), dim3(256),
                       0, s, x, sb, sc, st, w, bias, out, Cin, Cout, T, 0.f, AugCfg{});
    return hipGetLastError();
  }
  if (mean) {
    hipLaunchKernelGGL(conv1d_k3_bn_relu_kernel<true>, dim3(1, Cout / C1D_OT, B), dim3(256), 0, s, x, sb, sc, st, w,
                       bias, out, Cin, Cout, T, 1.0f / (float)T, AugCfg{});
  } else {
    hipLaunchKernelGGL(conv1d_k3_bn_relu_kernel<false>, dim3((T + C1D_TT - 1) / C1D_TT, Cout / C1D_OT, B), dim3(256), 0,
                       s, x, sb, sc, st, w, bias, out, Cin, Cout, T, 0.f, AugCfg{});
  }
  return hipGetLastError();
}

}  // namespace dfa
