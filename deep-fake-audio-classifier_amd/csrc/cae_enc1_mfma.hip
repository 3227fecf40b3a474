// cae_enc1_mfma.hip -- auto-encoder block 1 on the matrix cores (bf16 storage mode):
//   Conv2d(1, 32, 3, pad 1) + BatchNorm (folded) + ReLU + AvgPool2d(2)   (src/model_cae.py:34-37), FeatureNormalizer z-score fused
//   into the loads (src/dataset_cae.py:37-41).
// The vector-ALU kernel (cae.hip) spends 36 FMAs + 4 ReLUs per pooled output element: 4.2 G FMAs per 256 utterances, 0.157 ms at
// 35 % of the VALU peak, for 236 MB of output.  Here the 9-tap convolution is ONE K = 16 matrix product per pooled row and
// 32-column tile, the construction of conv12_fused.hip: K = 4 feature rows x (3 taps + a zero), the B operand of a lane is its
// column's window x[2q-1 .. 2q+2][f-1 .. f+1], and two A operands (weights on rows 0..2 / 1..3) give the even and the odd
// convolution row of the pool pair.  The features reach this kernel in fp32 (z-scored) and e1 is STORED in bf16: a product error
// of 2^-17 (two-term operands) moved 0.4 % of the stored e1 elements to the neighbouring bf16 value and 6 % of the latent
// (measured against the rounding-faithful oracle; the fp32 vector kernel: 2 %), so both operands are carried as THREE bf16 terms
// (hi + lo + lo2 = 24 bits) and a product is the six MFMAs whose weight is above 2^-25: every partial product is exact in the
// fp32 accumulator and only the accumulation rounds, as in the vector kernel -- 12 MFMAs per tile, still a fraction of its time.
// Epilogue in the accumulator layout (lane = column, 16 channels in registers): ReLU, vertical pair add (the pool's 1/2 rides on
// the packed weights), horizontal pair add with the neighbouring lane, bf16 pack, half-wave exchange -> every lane stores ONE
// 16-byte chunk of the pooled pixel (even columns the chunk of channels 8h.., odd columns the chunk 16 + 8h..).
// Workgroup = (utterance, 16 pooled rows, all columns): the 34 x (F + 2) feature rows go to LDS once (lanes along the
// contiguous axis of x, row pitch = 1 mod 32 floats: conflict-free stores either way); 4 waves share the 16 x ceil(F / 32) tiles.
#include "dfa_internal.h"
#include "conv3x3_mfma.h"
#include <stdlib.h>

namespace dfa {
namespace e1m {
constexpr int QG = 16, XR = 2 * QG + 2;      // pooled rows per workgroup; feature rows in LDS
}

// A operands [6][64]: even-{hi, lo, lo2}, odd-{hi, lo, lo2}; lane (channel lane & 31, half lane >> 5), element j <-> k = 8 h + j =
// 4 * (window row) + tap (tap 3 = zero); the even convolution row uses window rows 0..2, the odd one rows 1..3; the 2 x 2 pool's
// 1/4 is folded in (relu(s y) = s relu(y)).  w1 / b1: BatchNorm-folded fp32 weights [32][9] / bias [32].
__global__ void pack_cae_enc1_mfma_kernel(const float* __restrict__ w1, const float* __restrict__ b1, uint4* __restrict__ pack,
                                          float* __restrict__ bias) {
  const int i = threadIdx.x;                 // 384 threads: (operand i / 64, lane i % 64)
  if (i < 32) bias[i] = 0.25f * b1[i];
  const int op = i >> 6, lane = i & 63, ch = lane & 31, hh = lane >> 5;
  const bool odd = op >= 3;
  const int term = op % 3;
  bf16_t v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * hh + j, dyy = k >> 2, dx = k & 3;
    const int dy = odd ? dyy - 1 : dyy;
    float wv = 0.f;
    if (dx < 3 && dy >= 0 && dy <= 2) wv = 0.25f * w1[ch * 9 + dy * 3 + dx];
    const bf16_t t0 = float_to_bf16(wv);
    const float r1 = wv - bf16_to_float(t0);
    const bf16_t t1 = float_to_bf16(r1);
    const bf16_t t2 = float_to_bf16(r1 - bf16_to_float(t1));
    v[j] = term == 0 ? t0 : (term == 1 ? t1 : t2);
  }
  pack[i] = *reinterpret_cast<const uint4*>(v);
}
hipError_t launch_pack_cae_enc1_mfma(const float* w1, const float* b1, uint4* pack, float* bias, hipStream_t s) {
  hipLaunchKernelGGL(pack_cae_enc1_mfma_kernel, dim3(1), dim3(384), 0, s, w1, b1, pack, bias);
  return hipGetLastError();
}

template <typename TX>
__global__ __launch_bounds__(256) void cae_enc1_mfma_kernel(const TX* __restrict__ x, int64_t sb, int64_t st, int64_t sf,
                                                            const float* __restrict__ mu, const float* __restrict__ sigma,
                                                            const uint4* __restrict__ c1pack, const float* __restrict__ c1bias,
                                                            bf16_t* __restrict__ out, int T, int F, int Ho, int Wo, int pitch, int dbg) {
  using namespace e1m;
  extern __shared__ __attribute__((aligned(16))) float xs[];      // [XR][pitch]: column c <-> feature f = c - 1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, h = lane >> 5;
  const int b = blockIdx.y, q0 = blockIdx.x * QG;
  const TX* xb = x + (int64_t)b * sb;
  const int t_base = 2 * q0 - 1;
  const bool t_fast = (st == 1);
  const int ncol = F + 2;
  float* const zs = xs + XR * pitch;                  // [F][2]: 1 / sigma, -mu / sigma (z-score table, only when mu != null)
  if (mu) {
    for (int f = tid; f < F; f += 256) {
      const float rs = __builtin_amdgcn_rcpf(sigma[f]);
      zs[2 * f] = rs;
      zs[2 * f + 1] = -mu[f] * rs;
    }
    __syncthreads();
  }
  // Eight loads in flight per thread and trip (a one-load-per-trip loop pays the memory latency 25 times per workgroup: the
  // vector kernel it replaces does, and so did the first version of this one -- 0.158 ms for 0.06 ms of arithmetic).
  constexpr int NE = 8;
  const int nel = XR * ncol;
  for (int e0 = 0; e0 < nel; e0 += 256 * NE) {
    float v[NE];
    int dst[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
      const int e = e0 + k * 256 + tid;
      int rr, cc;
      if (t_fast) { cc = e / XR; rr = e - cc * XR; } else { rr = e / ncol; cc = e - rr * ncol; }
      const int t = t_base + rr, f = cc - 1;
      const bool ok = e < nel && t >= 0 && t < T && f >= 0 && f < F && !(dbg & 2);
      const TX* p = xb + (ok ? (int64_t)t * st + (int64_t)f * sf : 0);      // clamped address, branch-free
      float xv;
      if constexpr (sizeof(TX) == 2) xv = bf16_to_float(*p); else xv = *p;
      // z-score as x * (1 / sigma) - mu / sigma from a per-column table in LDS (built once per workgroup): one LDS read pair + one FMA
      // per element instead of two more global loads and an IEEE division (this loop is bound by its memory instructions); the 1-2
      // ulp difference disappears in the bf16 rounding of the operand (bf16 mode only: the fp32 parity path keeps the division)
      if (mu) { const int fc = ok ? f : 0; xv = fmaf(xv, zs[2 * fc], zs[2 * fc + 1]); }
      v[k] = ok ? xv : 0.f;                     // the convolution's zero padding applies to the NORMALISED input
      dst[k] = e < nel ? rr * pitch + cc : -1;
    }
#pragma unroll
    for (int k = 0; k < NE; ++k)
      if (dst[k] >= 0) xs[dst[k]] = v[k];
  }
  // A operands (even-hi, even-lo, odd-hi, odd-lo; conv12_fused.hip's pack: the vertical pool's 1/2 is folded in) and the bias as
  // the accumulators' initial value: channel (r & 3) + 8 (r >> 2) + 4 h <-> register r
  uint4 cw[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) cw[k] = c1pack[k * 64 + lane];
  f32x16_t bias;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 bv = *(const float4*)(c1bias + 8 * g + 4 * h);
    bias[4 * g] = bv.x; bias[4 * g + 1] = bv.y; bias[4 * g + 2] = bv.z; bias[4 * g + 3] = bv.w;
  }
  __syncthreads();

  const float rlim = relu_limit();
  const int ntile = (F + 31) / 32;
  const int nunit = min(QG, Ho - q0) * ntile;
  for (int u = wave; u < ((dbg & 4) ? 0 : nunit); u += 4) {
    const int qq = u / ntile, tile = u - qq * ntile;
    const int f = 32 * tile + col;
    // this lane's window rows 2 qq + 2 h, + 1 (of the 4-row window of pooled row q0 + qq), columns f - 1 .. f + 1 (zero beyond F:
    // the tile's columns past the image read the zeroed right-hand pad or the next row's pad -- clamp to the pad column)
    const int cc = min(f, F + 1 - 2);
    const float* r0 = xs + (2 * qq + 2 * h) * pitch + cc;
    float v[8];
    const bool inimg = f < F;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
#pragma unroll
      for (int k = 0; k < 3; ++k) v[4 * rr + k] = inimg ? r0[rr * pitch + k] : 0.f;
      v[4 * rr + 3] = 0.f;
    }
    unsigned hh[4], ll[4], l2[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      hh[p] = pack_bf16x2(v[2 * p], v[2 * p + 1]);
      const float ra = v[2 * p] - __uint_as_float(hh[p] << 16), rb = v[2 * p + 1] - __uint_as_float(hh[p] & 0xffff0000u);
      ll[p] = pack_bf16x2(ra, rb);
      l2[p] = pack_bf16x2(ra - __uint_as_float(ll[p] << 16), rb - __uint_as_float(ll[p] & 0xffff0000u));
    }
    const uint4 xh = make_uint4(hh[0], hh[1], hh[2], hh[3]), xl = make_uint4(ll[0], ll[1], ll[2], ll[3]),
                x2 = make_uint4(l2[0], l2[1], l2[2], l2[3]);
    // smallest terms first into the accumulator that starts from the bias
    f32x16_t e = Mma<bf16_t>::run(cw[0], x2, bias);
    f32x16_t o = Mma<bf16_t>::run(cw[3], x2, bias);
    e = Mma<bf16_t>::run(cw[2], xh, e);
    o = Mma<bf16_t>::run(cw[5], xh, o);
    e = Mma<bf16_t>::run(cw[1], xl, e);
    o = Mma<bf16_t>::run(cw[4], xl, o);
    e = Mma<bf16_t>::run(cw[0], xl, e);
    o = Mma<bf16_t>::run(cw[3], xl, o);
    e = Mma<bf16_t>::run(cw[1], xh, e);
    o = Mma<bf16_t>::run(cw[4], xh, o);
    e = Mma<bf16_t>::run(cw[0], xh, e);
    o = Mma<bf16_t>::run(cw[3], xh, o);
    // ReLU + 2 x 2 average: vertical pair in this lane (factor 1/2 in the weights), horizontal pair with the lane of column f ^ 1
    unsigned pk[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      float s0 = relu1(e[2 * p], rlim) + relu1(o[2 * p], rlim);
      float s1 = relu1(e[2 * p + 1], rlim) + relu1(o[2 * p + 1], rlim);
      // neighbouring column = neighbouring lane: a DPP quad_perm [1,0,3,2] move, not an LDS permute
      s0 += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s0), 0xB1, 0xF, 0xF, true));
      s1 += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(s1), 0xB1, 0xF, 0xF, true));
      pk[p] = pack_bf16x2(s0, s1);                             // (both pool factors ride on the packed weights and bias)
    }
    // half-wave exchange: lanes < 32 get channels 8 g .. 8 g + 7, lanes >= 32 channels 8 g + 8 .. 8 g + 15 (g = 0, 2)
    const auto a0 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
    const auto a1 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
    const auto b0 = __builtin_amdgcn_permlane32_swap(pk[4], pk[6], false, false);
    const auto b1 = __builtin_amdgcn_permlane32_swap(pk[5], pk[7], false, false);
    const int q = q0 + qq, j = f >> 1;
    const bool oddc = col & 1;                                 // even column: chunk g = 0 (+ h), odd column: chunk g = 2 (+ h)
    // (element-wise selects: indexing a two-element register array with the lane's parity sent it through scratch memory)
    const uint4 val = make_uint4(oddc ? b0[0] : a0[0], oddc ? b1[0] : a1[0], oddc ? b0[1] : a0[1], oddc ? b1[1] : a1[1]);
    if (j < Wo && !(dbg & 1))
      *(uint4*)((char*)(out + (((size_t)b * Ho + q) * Wo + j) * 32) + (2 * (int)oddc + h) * 16) = val;
  }
}

int cae_enc1_mfma_pitch(int F) { return (F + 2 + 31) / 32 * 32 + 1; }
size_t cae_enc1_mfma_lds(int F) { return ((size_t)e1m::XR * cae_enc1_mfma_pitch(F) + 2 * (size_t)F) * sizeof(float); }   // x rows + the z-score table

hipError_t launch_cae_enc1_mfma(const void* x, int x_dtype, int64_t sb, int64_t st, int64_t sf, const float* mu, const float* sigma,
                                const uint4* c1pack, const float* c1bias, void* out, int B, int T, int F, hipStream_t s) {
  const int Ho = T / 2, Wo = F / 2, pitch = cae_enc1_mfma_pitch(F);
  static const int dbg = getenv("DFA_E1_DBG") ? atoi(getenv("DFA_E1_DBG")) : 0;   // diagnostic phase skipping (timing only)
  const size_t lds = cae_enc1_mfma_lds(F);
  dim3 grid((Ho + e1m::QG - 1) / e1m::QG, B), block(256);
  if (x_dtype == DFA_DTYPE_BF16) {
    hipError_t e = hipFuncSetAttribute((const void*)cae_enc1_mfma_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cae_enc1_mfma_kernel<bf16_t>, grid, block, lds, s, (const bf16_t*)x, sb, st, sf, mu, sigma, c1pack, c1bias, (bf16_t*)out, T, F, Ho, Wo, pitch, dbg);
  } else {
    hipError_t e = hipFuncSetAttribute((const void*)cae_enc1_mfma_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(cae_enc1_mfma_kernel<float>, grid, block, lds, s, (const float*)x, sb, st, sf, mu, sigma, c1pack, c1bias, (bf16_t*)out, T, F, Ho, Wo, pitch, dbg);
  }
  return hipGetLastError();
}

}  // namespace dfa
