// pack.hip -- weight preparation kernels: fold eval-mode BatchNorm into the preceding convolution and
// lay the folded weights out in the order the forward kernels consume them.
//
// Folding (eval mode, src/predict.py:87): BN(conv(x)) = s*(conv_w*x) + (conv_b - mean)*s + beta with
// s = gamma / sqrt(var + eps)  (torch.nn.BatchNorm2d, src/model.py:16,22,28; eps = 1e-5).
#include "dfa_internal.h"

namespace dfa {

// block 1: w[32][1][3][3] -> w1[32][9] * s, b1[32]
__global__ void fold_conv1_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                  const float* __restrict__ g, const float* __restrict__ beta,
                                  const float* __restrict__ mean, const float* __restrict__ var,
                                  float* __restrict__ w1, float* __restrict__ b1, int cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cout * 9) {
    const int c = i / 9;
    const float s = g[c] / sqrtf(var[c] + kBnEps);
    w1[i] = w[i] * s;
  }
  if (i < cout) {
    const float s = g[i] / sqrtf(var[i] + kBnEps);
    b1[i] = (b[i] - mean[i]) * s + beta[i];
  }
}

// 3x3 conv, w[COUT][CIN][3][3] -> wpack[COUT/32][9][CIN/KG][64 lanes][16 bytes].
// Lane (r = lane&31, h = lane>>5), element j of its 16 bytes = s[co] * w[co = 32*slice + r][ci = KG*kg + (KG/2)*h + j][tap]
// which is exactly the B operand of v_mfma_f32_32x32x16_bf16 (KG = 16) / four v_mfma_f32_32x32x2_f32 (KG = 8).
template <typename T>
__global__ void fold_pack_conv3x3_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                         const float* __restrict__ g, const float* __restrict__ beta,
                                         const float* __restrict__ mean, const float* __restrict__ var,
                                         int cin_total, int cin_off, int cin, int cout, uint4* __restrict__ wpack,
                                         float* __restrict__ bias, int fold, float post_scale) {
  constexpr int KG = 32 / (int)sizeof(T);
  constexpr int EPL = KG / 2;  // elements per lane per k-group
  const int nkg = cin / KG;
  const int total = (cout / 32) * 9 * nkg * 64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cout) {
    // fold == 0: raw convolution (train mode: BN uses batch statistics and runs as its own pass)
    // post_scale: the average-pool factor (1/2, 1/4) of the fused epilogue, folded in: relu(s*x) = s*relu(x), s > 0
    const float s = fold ? g[i] / sqrtf(var[i] + kBnEps) : 1.f;
    bias[i] = (fold ? (b[i] - mean[i]) * s + beta[i] : b[i]) * post_scale;
  }
  if (i >= total) return;
  const int lane = i & 63;
  int rest = i >> 6;
  const int kg = rest % nkg;
  rest /= nkg;
  const int tap = rest % 9;
  const int slice = rest / 9;
  const int co = slice * 32 + (lane & 31), hh = lane >> 5;
  const float s = fold ? g[co] / sqrtf(var[co] + kBnEps) : 1.f;
  T v[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) {
    const int ci = KG * kg + EPL * hh + j;
    v[j] = cvt_out<T>(w[((size_t)co * cin_total + cin_off + ci) * 9 + tap] * s * post_scale);
  }
  wpack[i] = *reinterpret_cast<const uint4*>(v);
}

hipError_t launch_fold_conv1(const float* w, const float* b, const float* g, const float* beta, const float* mean,
                             const float* var, float* w1, float* b1, int cout, hipStream_t s) {
  const int n = cout * 9;
  hipLaunchKernelGGL(fold_conv1_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, b, g, beta, mean, var, w1, b1,
                     cout);
  return hipGetLastError();
}

hipError_t launch_fold_pack_conv3x3(const float* w, const float* b, const float* g, const float* beta,
                                    const float* mean, const float* var, int cin_total, int cin_off, int cin, int cout,
                                    int prec, uint4* wpack, float* bias, hipStream_t s, int fold, float post_scale) {
  const int kg = (prec == DFA_PREC_BF16) ? 16 : 8;
  int total = (cout / 32) * 9 * (cin / kg) * 64;
  if (total < cout) total = cout;
  dim3 grid((total + 255) / 256), block(256);
  if (prec == DFA_PREC_BF16)
    hipLaunchKernelGGL(fold_pack_conv3x3_kernel<bf16_t>, grid, block, 0, s, w, b, g, beta, mean, var, cin_total, cin_off,
                       cin, cout, wpack, bias, fold, post_scale);
  else
    hipLaunchKernelGGL(fold_pack_conv3x3_kernel<float>, grid, block, 0, s, w, b, g, beta, mean, var, cin_total, cin_off, cin,
                       cout, wpack, bias, fold, post_scale);
  return hipGetLastError();
}

// ConvTranspose2d k2 s2 (+ BN fold): w[CIN][COUT][2][2] -> wpack[4*COUT/32][CIN/KG][64][16 B]; GEMM column
// n = (2a+c)*COUT + co; lane (r,h) element j = s[co] * w[ci = KG*kg + (KG/2)*h + j][co][a][c].
template <typename T>
__global__ void fold_pack_convt2x2_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                          const float* __restrict__ g, const float* __restrict__ beta,
                                          const float* __restrict__ mean, const float* __restrict__ var, int cin,
                                          int cout, uint4* __restrict__ wpack, float* __restrict__ bias, int fold) {
  constexpr int KG = 32 / (int)sizeof(T);
  constexpr int EPL = KG / 2;
  const int nkg = cin / KG;
  const int total = (4 * cout / 32) * nkg * 64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cout) bias[i] = fold ? (b[i] - mean[i]) * (g[i] / sqrtf(var[i] + kBnEps)) + beta[i] : b[i];
  if (i >= total) return;
  const int lane = i & 63;
  int rest = i >> 6;
  const int kg = rest % nkg;
  const int slice = rest / nkg;
  const int n = slice * 32 + (lane & 31), hh = lane >> 5;
  const int q = n / cout, co = n - q * cout;
  const float s = fold ? g[co] / sqrtf(var[co] + kBnEps) : 1.f;
  T v[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) {
    const int ci = KG * kg + EPL * hh + j;
    v[j] = cvt_out<T>(w[((size_t)ci * cout + co) * 4 + q] * s);
  }
  wpack[i] = *reinterpret_cast<const uint4*>(v);
}

hipError_t launch_fold_pack_convt2x2(const float* w, const float* b, const float* g, const float* beta,
                                     const float* mean, const float* var, int cin, int cout, int prec, uint4* wpack,
                                     float* bias, hipStream_t s, int fold) {
  const int kg = (prec == DFA_PREC_BF16) ? 16 : 8;
  int total = (4 * cout / 32) * (cin / kg) * 64;
  if (total < cout) total = cout;
  dim3 grid((total + 255) / 256), block(256);
  if (prec == DFA_PREC_BF16)
    hipLaunchKernelGGL(fold_pack_convt2x2_kernel<bf16_t>, grid, block, 0, s, w, b, g, beta, mean, var, cin, cout, wpack,
                       bias, fold);
  else
    hipLaunchKernelGGL(fold_pack_convt2x2_kernel<float>, grid, block, 0, s, w, b, g, beta, mean, var, cin, cout, wpack,
                       bias, fold);
  return hipGetLastError();
}

// data-gradient ("dgrad") image of a 3x3 / pad 1 convolution: da = conv3x3(dz, W') with
//   W'[ci][co][dy'][dx'] = W[co][ci][2-dy'][2-dx']     (input channels = the forward conv's output channels).
// Packed like a forward conv with CIN' = cout window [co_off, co_off+co_n), COUT' = cin; no bias, no BN.
template <typename T>
__global__ void pack_conv3x3_dgrad_kernel(const float* __restrict__ w, int cin, int cout, int co_off, int co_n,
                                          uint4* __restrict__ wpack, float* __restrict__ bias) {
  constexpr int KG = 32 / (int)sizeof(T);
  constexpr int EPL = KG / 2;
  const int nkg = co_n / KG;
  const int total = (cin / 32) * 9 * nkg * 64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cin) bias[i] = 0.f;
  if (i >= total) return;
  const int lane = i & 63;
  int rest = i >> 6;
  const int kg = rest % nkg;
  rest /= nkg;
  const int tap = rest % 9;
  const int slice = rest / 9;
  const int ci = slice * 32 + (lane & 31), hh = lane >> 5;
  T v[EPL];
#pragma unroll
  for (int j = 0; j < EPL; ++j) {
    const int co = co_off + KG * kg + EPL * hh + j;
    v[j] = cvt_out<T>(w[((size_t)co * cin + ci) * 9 + (8 - tap)]);
  }
  wpack[i] = *reinterpret_cast<const uint4*>(v);
}

hipError_t launch_pack_conv3x3_dgrad(const float* w, int cin, int cout, int co_off, int co_n, int prec, uint4* wpack,
                                     float* bias, hipStream_t s) {
  const int kg = (prec == DFA_PREC_BF16) ? 16 : 8;
  int total = (cin / 32) * 9 * (co_n / kg) * 64;
  if (total < cin) total = cin;
  dim3 grid((total + 255) / 256), block(256);
  if (prec == DFA_PREC_BF16)
    hipLaunchKernelGGL(pack_conv3x3_dgrad_kernel<bf16_t>, grid, block, 0, s, w, cin, cout, co_off, co_n, wpack, bias);
  else
    hipLaunchKernelGGL(pack_conv3x3_dgrad_kernel<float>, grid, block, 0, s, w, cin, cout, co_off, co_n, wpack, bias);
  return hipGetLastError();
}

}  // namespace dfa
