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

typedef float f32x2_t __attribute__((ext_vector_type(2)));

template <typename TX>
__device__ __forceinline__ float load_x(const TX* p);
template <>
__device__ __forceinline__ float load_x<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float load_x<bf16_t>(const bf16_t* p) { return bf16_to_float(*p); }

// Thread = (pooled pixel, channel octet): lanes 4p..4p+3 own the four 8-channel groups of pixel p, so one wave store
// instruction writes 16 pixels x 64 bytes = 1 KiB of contiguous channels-last output.  The thread keeps its 8 x 9
// folded weights in registers (as 4 channel pairs x 9 taps) and walks 4 pixels of the 16 x 16 tile; every tap of every
// pre-pool row is one packed v_pk_fma_f32 on a channel pair.
// TO = split_t: the output pixel is [hi: 32 bf16][lo: 32 bf16] with v = hi + lo to 2^-17 (conv_split.hip consumes it)
struct split_t { unsigned short v; };

template <typename TX, typename TO>
__global__ __launch_bounds__(256) void conv1_bn_relu_poolh2_kernel(const TX* __restrict__ x, int64_t sb, int64_t st,
                                                                   int64_t sf, const float* __restrict__ w1,
                                                                   const float* __restrict__ b1, TO* __restrict__ out,
                                                                   int T, int F, int Ho, DropCfg dc, AugCfg aug) {
  __shared__ float xs[2][C1_XR][C1_XC + 1];
  const int tid = threadIdx.x;
  const int b = blockIdx.z;
  const int f0 = blockIdx.x * C1_TF;
  const TX* xb = x + (int64_t)b * sb;
  const int f_base = f0 - 1;
  const bool t_fast = (st == 1);
  constexpr int NX = (C1_XR * C1_XC + 255) / 256;   // x elements per thread per tile
  // x tile of the row tile starting at pooled row i0 -> registers (walk the contiguous axis with consecutive threads)
  auto load_tile = [&](int i0, TX* xr) {
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int e = k * 256 + tid;
      int rr, cc;
      if (t_fast) { cc = e / C1_XR; rr = e - cc * C1_XR; } else { rr = e / C1_XC; cc = e - rr * C1_XC; }
      const int t = 2 * i0 - 1 + rr, f = f_base + cc;
      // branch-free: always load from a clamped in-image address, zero afterwards -- the NX loads issue back to back
      const int tc = min(max(t, 0), T - 1), fc = min(max(f, 0), F - 1);
      xr[k] = xb[(int64_t)aug_src_t(aug, tc) * st + (int64_t)fc * sf];   // raw bits only: the first USE (and its wait) is in store_tile
    }
  };
  auto store_tile = [&](int buf, int i0, const TX* xr) {
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const int e = k * 256 + tid;
      int rr, cc;
      if (t_fast) { cc = e / C1_XR; rr = e - cc * C1_XR; } else { rr = e / C1_XC; cc = e - rr * C1_XC; }
      const int t = 2 * i0 - 1 + rr, f = f_base + cc;
      const bool ok = t >= 0 && t < T && f >= 0 && f < F;     // conv zero padding
      if (e < C1_XR * C1_XC) xs[buf][rr][cc] = ok ? aug_apply(aug, load_x<TX>(&xr[k]), b, t, f) : 0.f;   // train: augmentation folded in
    }
  };
  const int q = tid & 3, pl = tid >> 2;   // channel octet, pixel lane (0..63)
  f32x2_t wv[4][9], bv[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[c][k] = (f32x2_t){w1[(q * 8 + 2 * c) * 9 + k], w1[(q * 8 + 2 * c + 1) * 9 + k]};
    bv[c] = (f32x2_t){b1[q * 8 + 2 * c], b1[q * 8 + 2 * c + 1]};
  }
  const int fi = pl & (C1_TF - 1), rq = pl >> 4;   // 16 columns x 4 rows per pass
  const int f = f0 + fi;
  // the block walks row tiles blockIdx.y, blockIdx.y + gridDim.y, ...; the next tile's x is fetched (global -> VGPR)
  // while the current one is computed, and lands in the other LDS buffer before the single barrier of the iteration
  const int ntiles = (Ho + C1_TI - 1) / C1_TI;
  TX xr[NX];
  int buf = 0;
  if ((int)blockIdx.y < ntiles) {
    load_tile(blockIdx.y * C1_TI, xr);
    store_tile(0, blockIdx.y * C1_TI, xr);
  }
  __syncthreads();
  for (int tile = blockIdx.y; tile < ntiles; tile += gridDim.y) {
    const int i0 = tile * C1_TI;
    const bool more = tile + (int)gridDim.y < ntiles;
    if (more) load_tile((tile + gridDim.y) * C1_TI, xr);
    // the 4 passes keep their outputs in registers; the stores are issued after the prefetched x tile has been written to
    // LDS, so that wait never has this iteration's stores in front of it (vmcnt counts loads and stores in order)
    constexpr bool SPLIT = std::is_same<TO, split_t>::value;
    constexpr int NVS = SPLIT ? 2 : (int)(8 * sizeof(TO) / 16);
    uint4 outv[C1_TI / 4][NVS];
#pragma unroll
    for (int pass = 0; pass < C1_TI / 4; ++pass) {
      const int ri = pass * 4 + rq;
      f32x2_t xp[4][3];   // x rows 2i-1 .. 2i+2, broadcast to both halves of the channel pair
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int d = 0; d < 3; ++d) { const float xv = xs[buf][2 * ri + a][fi + d]; xp[a][d] = (f32x2_t){xv, xv}; }
      float o[8];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        f32x2_t a0 = bv[c], a1 = bv[c];   // pre-pool rows 2i and 2i+1
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int d = 0; d < 3; ++d) {
            a0 = __builtin_elementwise_fma(wv[c][dy * 3 + d], xp[dy][d], a0);
            a1 = __builtin_elementwise_fma(wv[c][dy * 3 + d], xp[dy + 1][d], a1);
          }
        o[2 * c] = 0.5f * (fmaxf(a0.x, 0.f) + fmaxf(a1.x, 0.f));
        o[2 * c + 1] = 0.5f * (fmaxf(a0.y, 0.f) + fmaxf(a1.y, 0.f));
      }
      if (dc.thresh != 0) {
        const int i = i0 + ri;
        float ds[8];
        drop_scale8(dc, (((uint64_t)b * Ho + i) * F + f) * 32 + q * 8, ds);
#pragma unroll
        for (int c = 0; c < 8; ++c) o[c] *= ds[c];
      }
      if constexpr (SPLIT) {
        float h[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) h[c] = bf16_to_float(float_to_bf16(o[c]));
        outv[pass][0] = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
        outv[pass][1] = make_uint4(pack_bf16x2(o[0] - h[0], o[1] - h[1]), pack_bf16x2(o[2] - h[2], o[3] - h[3]),
                                   pack_bf16x2(o[4] - h[4], o[5] - h[5]), pack_bf16x2(o[6] - h[6], o[7] - h[7]));
      } else {
        TO ov[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) ov[c] = cvt_out<TO>(o[c]);
#pragma unroll
        for (int v = 0; v < NVS; ++v) outv[pass][v] = reinterpret_cast<const uint4*>(ov)[v];
      }
    }
    if (more) store_tile(buf ^ 1, (tile + gridDim.y) * C1_TI, xr);
#pragma unroll
    for (int pass = 0; pass < C1_TI / 4; ++pass) {
      const int i = i0 + pass * 4 + rq;
      if (i < Ho && f < F) {
        if constexpr (SPLIT) {
          unsigned short* op = (unsigned short*)out + (((uint64_t)b * Ho + i) * F + f) * 64 + q * 8;
          *reinterpret_cast<uint4*>(op) = outv[pass][0];          // hi plane: channels 8q .. 8q+7
          *reinterpret_cast<uint4*>(op + 32) = outv[pass][1];     // lo plane
        } else {
          TO* op = out + (((uint64_t)b * Ho + i) * F + f) * 32 + q * 8;
#pragma unroll
          for (int v = 0; v < NVS; ++v) reinterpret_cast<uint4*>(op)[v] = outv[pass][v];
        }
      }
    }
    __syncthreads();
    buf ^= 1;
  }
}

hipError_t launch_conv1(const void* x, int x_dtype, int64_t sb, int64_t st, int64_t sf, const float* w1,
                        const float* b1, void* out, int out_prec, int B, int T, int F, hipStream_t s,
                        const DropCfg* drop, const AugCfg* augp) {
  DropCfg dc{};
  if (drop) dc = *drop;
  AugCfg aug{};
  if (augp) aug = *augp;
  const int Ho = T / 2;
  // few row-tile walkers per (utterance, column tile) once the batch alone fills the chip; more for small batches
  const int ntiles = (Ho + C1_TI - 1) / C1_TI, ncols = (F + C1_TF - 1) / C1_TF;
  int ny = (4096 + B * ncols - 1) / (B * ncols);
  ny = ny < 1 ? 1 : (ny > ntiles ? ntiles : ny);
  dim3 grid(ncols, ny, B), block(256);
  if (out_prec == DFA_PREC_BF16X3) {
    if (x_dtype == DFA_DTYPE_F32)
      hipLaunchKernelGGL((conv1_bn_relu_poolh2_kernel<float, split_t>), grid, block, 0, s, (const float*)x, sb, st, sf, w1,
                         b1, (split_t*)out, T, F, Ho, dc, aug);
    else
      hipLaunchKernelGGL((conv1_bn_relu_poolh2_kernel<bf16_t, split_t>), grid, block, 0, s, (const bf16_t*)x, sb, st, sf,
                         w1, b1, (split_t*)out, T, F, Ho, dc, aug);
    return hipGetLastError();
  }
  if (x_dtype == DFA_DTYPE_F32 && out_prec == DFA_PREC_F32)
    hipLaunchKernelGGL((conv1_bn_relu_poolh2_kernel<float, float>), grid, block, 0, s, (const float*)x, sb, st, sf, w1,
                       b1, (float*)out, T, F, Ho, dc, aug);
  else if (x_dtype == DFA_DTYPE_F32 && out_prec == DFA_PREC_BF16)
    hipLaunchKernelGGL((conv1_bn_relu_poolh2_kernel<float, bf16_t>), grid, block, 0, s, (const float*)x, sb, st, sf,
                       w1, b1, (bf16_t*)out, T, F, Ho, dc, aug);
  else if (x_dtype == DFA_DTYPE_BF16 && out_prec == DFA_PREC_F32)
    hipLaunchKernelGGL((conv1_bn_relu_poolh2_kernel<bf16_t, float>), grid, block, 0, s, (const bf16_t*)x, sb, st, sf,
                       w1, b1, (float*)out, T, F, Ho, dc, aug);
  else
    hipLaunchKernelGGL((conv1_bn_relu_poolh2_kernel<bf16_t, bf16_t>), grid, block, 0, s, (const bf16_t*)x, sb, st, sf,
                       w1, b1, (bf16_t*)out, T, F, Ho, dc, aug);
  return hipGetLastError();
}

}  // namespace dfa
