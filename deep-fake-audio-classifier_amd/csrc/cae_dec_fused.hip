// cae_dec_fused.hip -- the auto-encoder's whole decoder + reconstruction error as ONE kernel (bf16 storage mode):
//   ConvT 256->128 + BN + ReLU -> ConvT 128->64 (output_padding (0,1)) + BN + ReLU -> ConvT 64->32 + BN + ReLU -> ConvT 32->1
//   -> zero-pad T -> per-sample mean((recon - x)^2)        (src/model_cae.py:57-80,107-125; src/evaluation_cae.py:52-53).
//
// ConvTranspose2d(kernel 2, stride 2) never overlaps: a latent pixel owns its 2x2 -> 4x4 -> 8x8 -> 16x16 patch, so the three
// intermediates (d1, d2, d3: 3.2 MB per utterance written and re-read by the four-launch path, 0.55 ms of its 1.25 ms) need never
// leave the CU.  A workgroup takes 32 consecutive latent pixels of one utterance (16 KB, contiguous: channels-last) and runs the
// chain with the activations in LDS:
//   phase A  d1[128 px][128] = relu(W1 . lat)     M = 512 (q, co) x N = 32  x K = 256    256 MFMAs
//   phase B  d2[512 px][64]  = relu(W2 . d1)      M = 256         x N = 128 x K = 128    256 MFMAs
//   phase C  d3 = relu(W3 . d2) in REGISTERS      M = 128         x N = 512 x K = 64     256 MFMAs, then the 32 -> 1 layer as 64 FMAs
//            per lane on its 16 channels + one half-wave exchange, the z-scored x read through the caller's strides, and the
//            squared error -- no d3, no reconstruction in memory (recon is written only when the caller asks for it).
// v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the A operand (the packed images of convt2x2_mfma.h serve unchanged: the register
// image of a B-operand column block is that of an A-operand row block) and pixels as columns: a lane then holds 4 consecutive
// channels x 4 groups of one pixel, so outputs leave as 8-byte LDS stores (the four-launch kernels store single bf16 elements).
// Pixel order is the quadtree order (child = 4 * parent + q): only phase C maps a pixel back to (t, f).  Rounding points are those
// of the four-launch path (d1, d2, d3 rounded to bf16, fp32 accumulation), so the emulated-oracle tests apply unchanged.
// The output_padding column of block 2 is a per-channel constant, and so is everything grown from it: reconstruction columns
// 16 W4 .. 16 W4 + 3 are a 4 x 4 pattern of constants (row mod 4, column) computed once at prepare time (cae_opad_consts_kernel);
// the last workgroup of an utterance adds their error and that of the zero rows t >= 16 H4.
#include "dfa_internal.h"
#include "convt2x2_mfma.h"

namespace dfa {

struct CaeDecFusedArgs {
  const bf16_t* lat;               // [B][H4*W4][256]
  const uint4 *wp1, *wp2, *wp3;    // [4*COUT/32][CIN/16][64] x 16 bytes (launch_fold_pack_convt2x2, bf16)
  const float *b1, *b2, *b3;       // folded biases [128], [64], [32]
  const float *w4, *b4;            // ConvTranspose2d(32 -> 1) weight [32][4], bias [1] (raw, fp32)
  const uint4* w4pack;             // the same weights as MFMA A operands [2 k-steps][hi, lo][64] (pack_cae_dec4_kernel)
  const float* cst;                // [16] reconstruction constants of the output_padding columns: cst[(t & 3) * 4 + (f - 16 W4)]
  const void* x;
  int x_bf16;
  int64_t sb, st, sf;
  const float *mu, *sigma;         // fused FeatureNormalizer z-score, or null
  float* recon;                    // [B][T][F] or null
  float* partial;                  // [B][ntile] squared-error sums
  int H4, W4, T, F, ntile;
  long long* stamps;               // diagnostic (context option "clock_probe"): s_memtime at the phase boundaries, workgroups (tile, b < 18)
};

namespace cdf {
#ifndef DFA_CDF_NP
#define DFA_CDF_NP 64
#endif
constexpr int NP = DFA_CDF_NP;               // latent pixels per workgroup (32 or 64)
constexpr int LAT_B = NP * 512, D1_B = 4 * NP * 256;
constexpr int B2_OFF = LAT_B + D1_B;         // [64] float: block 2's folded bias (read per unit; global loads there would serialise)
constexpr int W3_OFF = B2_OFF + 256;         // [4 q3][4 k-steps][64 lanes] x 16 B: block 3's fragments in the permuted channel order
constexpr int RED_OFF = W3_OFF + 16384;      // [8] float
constexpr int ZS_OFF = RED_OFF + 64;          // [F <= 1024][2] float: 1 / sigma, -mu / sigma (z-score table of the NORM instantiations)
constexpr int ZS_MAXF = 1024;
constexpr int LDS_BYTES = ZS_OFF + 2 * ZS_MAXF * 4;
}  // namespace cdf

__device__ __forceinline__ float cdf_ldx(const CaeDecFusedArgs& a, int b, int t, int f) {
  const int64_t off = (int64_t)b * a.sb + (int64_t)t * a.st + (int64_t)f * a.sf;
  float v = a.x_bf16 ? bf16_to_float(((const bf16_t*)a.x)[off]) : ((const float*)a.x)[off];
  if (a.mu) v = (v - a.mu[f]) / a.sigma[f];
  return v;
}

// raw x element (no branches: XBF is a template parameter, so eight of these issue back to back)
template <bool XBF>
__device__ __forceinline__ float cdf_ldraw(const void* xu, unsigned off) {   // xu = the utterance's base (uniform), off in elements
  if constexpr (XBF) return bf16_to_float(((const bf16_t*)xu)[off]);
  else return ((const float*)xu)[off];
}

template <bool XBF, bool NORM>
__global__ __launch_bounds__(512, 1) void cae_dec_fused_kernel(CaeDecFusedArgs a) {
  using namespace cdf;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const latS = smem;
  char* const d1S = smem + LAT_B;
  float* const red = (float*)(smem + RED_OFF);
  const float* const zsS = (const float*)(smem + ZS_OFF);
  float* const b2S = (float*)(smem + B2_OFF);
  const uint4* const w3S = (const uint4*)(smem + W3_OFF);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, h = lane >> 5;
  const int tile = blockIdx.x, b = blockIdx.y;
  const int npx = a.H4 * a.W4, g0 = tile * NP;
  const float rlim = relu_limit();
  const int sid = b * a.ntile + tile;
  const bool stamp = a.stamps != nullptr && tid == 0 && sid < 128;
  if (stamp) { a.stamps[8 * sid] = __builtin_amdgcn_s_memtime(); a.stamps[8 * sid + 6] = __builtin_amdgcn_s_memrealtime(); }

  uint4 wa[16], wn[16];            // phase A's weight fragments (m-tiles 2 wave, 2 wave + 1), requested inside the staging block
  // ---- stage the latent tile (NP pixels x 512 B, contiguous) with the chunk swizzle of the fragment reads; b2 and W3 -> LDS.
  //      All loads first and unconditional (clamped address, masked value): a load under a branch costs a full wait each.
  {
    const char* src = (const char*)(a.lat + ((size_t)b * npx + g0) * 256);
    const int nvalid = (npx - g0) * 32;              // 16-byte chunks of this tile that exist
    uint4 v[NP / 16];
#pragma unroll
    for (int it = 0; it < NP / 16; ++it) {
      const int g = tid + 512 * it;
      v[it] = *(const uint4*)(src + (size_t)(g < nvalid ? g : 0) * 16);
    }
    // W3 fragment of lane (i, hh) in the order the d2 accumulators present their channels: elements 0-3 = the standard image's
    // lane (i, 0) elements 4 hh .. 4 hh + 3, elements 4-7 = lane (i, 1)'s
    const uint2* w3h = reinterpret_cast<const uint2*>(a.wp3);
    uint2 e[2][2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int f = (tid >> 6) + 8 * it;
      e[it][0] = w3h[((size_t)f * 64 + col) * 2 + h];
      e[it][1] = w3h[((size_t)f * 64 + 32 + col) * 2 + h];
    }
    const float b2v = a.b2[tid & 63];
    // phase A's fragments go out behind the tile's loads (memory returns in order: the LDS writes below wait for the tile only,
    // the 32 fragment loads stay in flight across the barrier)
    __builtin_amdgcn_sched_barrier(0);
    {
      const uint4* wp = a.wp1 + lane;
#pragma unroll
      for (int kg = 0; kg < 16; ++kg) wa[kg] = wp[(size_t)((2 * wave) * 16 + kg) * 64];
#pragma unroll
      for (int kg = 0; kg < 16; ++kg) wn[kg] = wp[(size_t)((2 * wave + 1) * 16 + kg) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < NP / 16; ++it) {
      const int g = tid + 512 * it, p = g >> 5, c = g & 31;
      const bool ok = g < nvalid;
      *(uint4*)(latS + p * 512 + ((c ^ (p & 15)) << 4)) = make_uint4(ok ? v[it].x : 0u, ok ? v[it].y : 0u, ok ? v[it].z : 0u, ok ? v[it].w : 0u);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it)
      *(uint4*)(smem + W3_OFF + ((((tid >> 6) + 8 * it) * 64 + lane) << 4)) = make_uint4(e[it][0].x, e[it][0].y, e[it][1].x, e[it][1].y);
    if (tid < 64) b2S[tid] = b2v;
    if constexpr (NORM) {                          // z-score table: x_hat = x * zs[f][0] + zs[f][1]
      float* zs = (float*)(smem + ZS_OFF);
      for (int f = tid; f < a.F; f += 512) {
        const float rs = __builtin_amdgcn_rcpf(a.sigma[f]);   // (v_rcp_f32: the error term is fp32, 1-2 ulp are far below the 2e-5 score tolerance)
        zs[2 * f] = rs;
        zs[2 * f + 1] = -a.mu[f] * rs;
      }
    }
  }
  __syncthreads();
  if (stamp) a.stamps[8 * sid + 1] = __builtin_amdgcn_s_memtime();

  // bias + ReLU + bf16 of one accumulator, 4 consecutive channels per 8-byte store at pixel P (row pitch PB, swizzle SW)
  auto store_tile = [&](const f32x16_t& acc, const float* bias, int co_base, char* dst, int P, int PB, int sw) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co = co_base + 8 * g + 4 * h;
      const float4 bv = *(const float4*)(bias + co);
      const unsigned lo = pack_bf16x2(relu1(acc[4 * g] + bv.x, rlim), relu1(acc[4 * g + 1] + bv.y, rlim));
      const unsigned hi = pack_bf16x2(relu1(acc[4 * g + 2] + bv.z, rlim), relu1(acc[4 * g + 3] + bv.w, rlim));
      *(uint2*)(dst + P * PB + ((((co >> 3)) ^ sw) << 4) + 8 * h) = make_uint2(lo, hi);
    }
  };

  const int q2 = wave & 3;
  uint4 w2[2][8];                  // phases B + C: block 2's m-tiles 2 q2, 2 q2 + 1 (requested inside phase A, see there)
  // ---- phase A: d1 = relu(W1 . lat): wave owns m-tiles 2 wave, 2 wave + 1 (n = q1 * 128 + co)
  {
    const int sw = col & 15;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int mt = 2 * wave + mi;
      f32x16_t acc[NP / 32];
#pragma unroll
      for (int n = 0; n < NP / 32; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
      for (int kg = 0; kg < 16; ++kg)
#pragma unroll
        for (int n = 0; n < NP / 32; ++n) {
          const uint4 xv = *(const uint4*)(latS + (32 * n + col) * 512 + (((2 * kg + h) ^ sw) << 4));
          acc[n] = Mma<bf16_t>::run(mi == 0 ? wa[kg] : wn[kg], xv, acc[n]);
        }
      // this m-tile's 16 fragments are dead from here: half of block 2's take their place while the stores / the next m-tile run
      __builtin_amdgcn_sched_barrier(0);   // (not earlier: the fragment registers of three layers at once would spill)
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) w2[mi][kg] = a.wp2[(size_t)((2 * q2 + mi) * 8 + kg) * 64 + lane];
#pragma unroll
      for (int n = 0; n < NP / 32; ++n) {
        const int P1 = 4 * (32 * n + col) + (mt >> 2);
        store_tile(acc[n], a.b1, 32 * (mt & 3), d1S, P1, 256, P1 & 15);
      }
    }
  }
  __syncthreads();
  if (stamp) a.stamps[8 * sid + 2] = __builtin_amdgcn_s_memtime();

  // ---- phases B + C, one register chain per wave, no LDS and no barrier between them.  A unit = (32 d1 pixels, q2): the wave
  //      computes BOTH 32-channel halves of d2 for those pixels' q2 children (m-tiles 2 q2, 2 q2 + 1), so their 64 channels sit in
  //      its own two accumulators; converted pairwise to bf16 they are the four k-steps of block 3's B operand ("accumulator as
  //      the next operand": registers 8 s .. 8 s + 7 of half hf -> k-step 2 hf + s, channel order 8 (j >> 2) + 4 h + (j & 3)
  //      inside a step -- W3's fragments are loaded in that same permuted order), d3 likewise feeds the 32 -> 1 layer.
  float err = 0.f;
  {
    f32x16_t b3v;                                   // block 3's bias in accumulator layout: the C operand of a chain's first MFMA
#pragma unroll
    for (int r = 0; r < 16; ++r) b3v[r] = a.b3[(r & 3) + 8 * (r >> 2) + 4 * h];
    uint4 w4f[4];                                   // [k-step 0: hi, lo][k-step 1: hi, lo]
#pragma unroll
    for (int i = 0; i < 4; ++i) w4f[i] = a.w4pack[i * 64 + lane];
    const float b4 = a.b4[0];
    const void* const xu = (const char*)a.x + (int64_t)b * a.sb * (XBF ? 2 : 4);
    const unsigned ust = (unsigned)a.st, usf = (unsigned)a.sf;
#pragma unroll 1
    for (int u = 0; u < NP / 16; ++u) {
      const int nt = (wave >> 2) + 2 * u;
      const int P1 = 32 * nt + col;
      const int p = P1 >> 2, q1 = P1 & 3;
      const bool valid = g0 + p < npx;
      const int g = valid ? g0 + p : npx - 1;        // clamped: the loads below are unconditional, the error is masked
      const int i4 = g / a.W4, j4 = g - i4 * a.W4;
      const int tq = 16 * i4 + 8 * (q1 >> 1) + 4 * (q2 >> 1) + h, fq = 16 * j4 + 8 * (q1 & 1) + 4 * (q2 & 1);
      // this lane's eight x values of the unit (rows tq, tq + 2; columns fq .. fq + 3), requested before the MFMA chain starts
      float xr[4][2], mu4[4], sg4[4];
      {
        const unsigned o0 = (unsigned)tq * ust + (unsigned)fq * usf;   // 32-bit inside an utterance (cae_dec_fused_supports)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int c = 0; c < 2; ++c) xr[mt][c] = cdf_ldraw<XBF>(xu, o0 + 2u * (mt >> 1) * ust + (2u * (mt & 1) + c) * usf);
        if constexpr (NORM) {
#pragma unroll
          for (int c = 0; c < 4; ++c) { sg4[c] = zsS[2 * (fq + c)]; mu4[c] = zsS[2 * (fq + c) + 1]; }      // LDS table (global loads here would wait)
        }
      }
      uint4 dk[4];
      {
        const char* xb = d1S + P1 * 256;
        const int sw = P1 & 15;
        f32x16_t acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll
        for (int kg = 0; kg < 8; ++kg) {
          const uint4 xv = *(const uint4*)(xb + (((2 * kg + h) ^ sw) << 4));
          acc0 = Mma<bf16_t>::run(w2[0][kg], xv, acc0);
          acc1 = Mma<bf16_t>::run(w2[1][kg], xv, acc1);
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int sI = 0; sI < 2; ++sI) {
            unsigned pk[4];
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
              const float4 bv = *(const float4*)(b2S + 32 * hf + 16 * sI + 8 * gq + 4 * h);
              const int r0 = 8 * sI + 4 * gq;
              const f32x16_t& ac = hf ? acc1 : acc0;
              pk[2 * gq] = pack_bf16x2(relu1(ac[r0] + bv.x, rlim), relu1(ac[r0 + 1] + bv.y, rlim));
              pk[2 * gq + 1] = pack_bf16x2(relu1(ac[r0 + 2] + bv.z, rlim), relu1(ac[r0 + 3] + bv.w, rlim));
            }
            dk[2 * hf + sI] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
          }
      }
      // W3 fragments come from LDS one m-tile ahead (all sixteen hoisted to the top of the unit would cost 64 registers and spill)
      uint4 w3f[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) w3f[kk] = w3S[kk * 64 + lane];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        __builtin_amdgcn_sched_barrier(0);
        f32x16_t acc = Mma<bf16_t>::run(w3f[0], dk[0], b3v);
#pragma unroll
        for (int kk = 1; kk < 4; ++kk) acc = Mma<bf16_t>::run(w3f[kk], dk[kk], acc);
        __builtin_amdgcn_sched_barrier(0);
        if (mt < 3) {
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) w3f[kk] = w3S[((mt + 1) * 4 + kk) * 64 + lane];
        }
        uint4 dfr[2];
#pragma unroll
        for (int sI = 0; sI < 2; ++sI) {
          unsigned pk[4];
#pragma unroll
          for (int pp = 0; pp < 4; ++pp)
            pk[pp] = pack_bf16x2(relu1(acc[8 * sI + 2 * pp], rlim), relu1(acc[8 * sI + 2 * pp + 1], rlim));
          dfr[sI] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
        f32x16_t y;
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = 0.f;
        y = Mma<bf16_t>::run(w4f[1], dfr[0], y);
        y = Mma<bf16_t>::run(w4f[3], dfr[1], y);
        y = Mma<bf16_t>::run(w4f[0], dfr[0], y);
        y = Mma<bf16_t>::run(w4f[2], dfr[1], y);
        // rows 4 h, 4 h + 1 of y = outputs (a4 = h, c4 = 0, 1) of this lane's pixel: registers 0, 1
        const float r0 = y[0] + b4, r1 = y[1] + b4;
        if (valid) {
          float x0 = xr[mt][0], x1 = xr[mt][1];
          if constexpr (NORM) { x0 = fmaf(x0, sg4[2 * (mt & 1)], mu4[2 * (mt & 1)]); x1 = fmaf(x1, sg4[2 * (mt & 1) + 1], mu4[2 * (mt & 1) + 1]); }
          const float d0 = r0 - x0, d1 = r1 - x1;
          err = fmaf(d0, d0, err);
          err = fmaf(d1, d1, err);
          if (a.recon) *reinterpret_cast<float2*>(a.recon + ((size_t)b * a.T + tq + 2 * (mt >> 1)) * a.F + fq + 2 * (mt & 1)) = make_float2(r0, r1);
        }
      }
    }
  }
  if (stamp) a.stamps[8 * sid + 3] = __builtin_amdgcn_s_memtime();
  // ---- the output_padding columns (constants) and the zero rows t >= 16 H4: the utterance's last workgroup
  if (tile == a.ntile - 1) {
    const int HR = 16 * a.H4, f0 = 16 * a.W4, nstrip = HR * 4, ntail = (a.T - HR) * a.F;
    for (int i = tid; i < nstrip + ntail; i += 512) {
      int t, f;
      float r;
      if (i < nstrip) { t = i >> 2; f = f0 + (i & 3); r = a.cst[(t & 3) * 4 + (i & 3)]; }
      else { const int k = i - nstrip; t = HR + k / a.F; f = k - (t - HR) * a.F; r = 0.f; }
      const float d = r - cdf_ldx(a, b, t, f);
      err = fmaf(d, d, err);
      if (a.recon) a.recon[((size_t)b * a.T + t) * a.F + f] = r;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) err += __shfl_down(err, off, 64);
  if (lane == 0) red[wave] = err;
  __syncthreads();
  if (stamp) { a.stamps[8 * sid + 4] = __builtin_amdgcn_s_memtime(); a.stamps[8 * sid + 7] = __builtin_amdgcn_s_memrealtime(); }
  if (tid == 0)
    a.partial[(size_t)b * a.ntile + tile] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
}

// Reconstruction values of the columns grown from block 2's output_padding column (a per-channel constant):
//   c2[ci] = bf16(relu(b2[ci]));  c3[q3][co] = bf16(relu(b3[co] + sum_ci W3[ci][q3, co] c2[ci]))  (W3 = the bf16 MFMA image);
//   cst[(2 a3 + a4) * 4 + 2 c3 + c4] = b4 + sum_co c3[(a3, c3)][co] * W4[co][(a4, c4)].
__global__ __launch_bounds__(128) void cae_opad_consts_kernel(const float* __restrict__ b2, const uint4* __restrict__ wp3,
                                                              const float* __restrict__ b3, const float* __restrict__ w4,
                                                              const float* __restrict__ b4, float* __restrict__ cst) {
  __shared__ float c2[64], c3[4][32];
  const int tid = threadIdx.x;
  if (tid < 64) c2[tid] = bf16_to_float(float_to_bf16(fmaxf(b2[tid], 0.f)));
  __syncthreads();
  {
    const int q3 = tid >> 5, co = tid & 31;
    float s = b3[co];
    for (int ci = 0; ci < 64; ++ci) {
      const int kg = ci >> 4, hh = (ci >> 3) & 1, j = ci & 7;
      const unsigned short* frag = (const unsigned short*)(wp3 + (size_t)(q3 * 4 + kg) * 64 + co + 32 * hh);
      bf16_t w;
      w.v = frag[j];
      s = fmaf(bf16_to_float(w), c2[ci], s);
    }
    c3[q3][co] = bf16_to_float(float_to_bf16(fmaxf(s, 0.f)));
  }
  __syncthreads();
  if (tid < 16) {
    const int q3 = tid >> 2, q4 = tid & 3;
    float s = b4[0];
    for (int co = 0; co < 32; ++co) s = fmaf(c3[q3][co], w4[co * 4 + q4], s);
    const int trow = 2 * (q3 >> 1) + (q4 >> 1), fcol = 2 * (q3 & 1) + (q4 & 1);
    cst[trow * 4 + fcol] = s;
  }
}

// W4 [32 ch][4] as the A operand of v_mfma_f32_32x32x16_bf16 for "accumulator tile as the next operand": rows 0, 1 = outputs
// q4 = 0, 1 and rows 4, 5 = outputs 2, 3 (so that lane half h finds its patch row a4 = h in accumulator registers 0, 1), zero
// elsewhere; k-step s, lane half hh, element j <-> channel 16 s + 8 (j >> 2) + 4 hh + (j & 3) -- the order in which registers
// 8 s .. 8 s + 7 of a 32 x 32 accumulator hold their rows (cdna_hip_programming.md section 3).  pack[2 s + part][lane], part 0 = hi.
__global__ void pack_cae_dec4_kernel(const float* __restrict__ w4, uint4* __restrict__ pack) {
  const int i = threadIdx.x;                     // 256 = [s][part][lane]
  const int lane = i & 63, part = (i >> 6) & 1, s = i >> 7;
  const int row = lane & 31, hh = lane >> 5;
  const int q4 = row == 0 ? 0 : row == 1 ? 1 : row == 4 ? 2 : row == 5 ? 3 : -1;
  bf16_t v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 16 * s + 8 * (j >> 2) + 4 * hh + (j & 3);
    const float w = q4 >= 0 ? w4[c * 4 + q4] : 0.f;
    const bf16_t hi = float_to_bf16(w);
    v[j] = part ? float_to_bf16(w - bf16_to_float(hi)) : hi;
  }
  pack[i] = *reinterpret_cast<const uint4*>(v);
}
hipError_t launch_pack_cae_dec4(const float* w4, uint4* pack, hipStream_t s) {
  hipLaunchKernelGGL(pack_cae_dec4_kernel, dim3(1), dim3(256), 0, s, w4, pack);
  return hipGetLastError();
}

hipError_t launch_cae_opad_consts(const float* b2, const uint4* wp3, const float* b3, const float* w4, const float* b4, float* cst,
                                  hipStream_t s) {
  hipLaunchKernelGGL(cae_opad_consts_kernel, dim3(1), dim3(128), 0, s, b2, wp3, b3, w4, b4, cst);
  return hipGetLastError();
}

// the kernel addresses x inside one utterance with unsigned 32-bit element offsets
bool cae_dec_fused_supports(int T, int F, int64_t st, int64_t sf) {
  return st >= 0 && sf >= 0 && F <= cdf::ZS_MAXF && (int64_t)(T - 1) * st + (int64_t)(F - 1) * sf < ((int64_t)1 << 31);
}

int cae_dec_fused_tiles(int H4, int W4) { return (H4 * W4 + cdf::NP - 1) / cdf::NP; }

hipError_t launch_cae_dec_fused(const void* lat, const uint4* wp1, const float* b1, const uint4* wp2, const float* b2, const uint4* wp3,
                                const float* b3, const float* w4, const float* b4, const uint4* w4pack, const float* cst, const void* x,
                                int x_dtype, int64_t sb, int64_t st, int64_t sf, const float* mu, const float* sigma, float* recon,
                                float* partial, int B, int H4, int W4, int T, int F, hipStream_t s, long long* stamps) {
  CaeDecFusedArgs a{};
  a.w4pack = w4pack;
  a.stamps = stamps;
  a.lat = (const bf16_t*)lat; a.wp1 = wp1; a.wp2 = wp2; a.wp3 = wp3; a.b1 = b1; a.b2 = b2; a.b3 = b3; a.w4 = w4; a.b4 = b4; a.cst = cst;
  a.x = x; a.x_bf16 = x_dtype == DFA_DTYPE_BF16 ? 1 : 0; a.sb = sb; a.st = st; a.sf = sf; a.mu = mu; a.sigma = sigma;
  a.recon = recon; a.partial = partial; a.H4 = H4; a.W4 = W4; a.T = T; a.F = F; a.ntile = cae_dec_fused_tiles(H4, W4);
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, cdf::LDS_BYTES);
    if (e != hipSuccess) return e;     // (per device: set on every launch, it is cheap)
    hipLaunchKernelGGL(kern, dim3(a.ntile, B), dim3(512), cdf::LDS_BYTES, s, a);
    return hipGetLastError();
  };
  if (a.x_bf16) return mu ? go(cae_dec_fused_kernel<true, true>) : go(cae_dec_fused_kernel<true, false>);
  return mu ? go(cae_dec_fused_kernel<false, true>) : go(cae_dec_fused_kernel<false, false>);
}

}  // namespace dfa
