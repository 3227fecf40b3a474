// cnn1d_fused_x3.hip -- the CNN1D eval forward (src/model_cnn1d.py:37-46) as ONE kernel on the bf16 matrix cores at fp32-grade
// accuracy: every weight and activation is carried as hi + lo bf16 (16 significant bits), every product is three
// v_mfma_f32_32x32x16_bf16 (hi*hi + lo*hi + hi*lo, fp32 accumulate; the dropped lo*lo term is 2^-18 of a product) -- the
// DFA_PREC_BF16X3 construction of conv_split.hip.  The exact-fp32 form (cnn1d_fused.hip, v_mfma_f32_32x32x2_f32) is held at
// ~80 cycles per 2048 MACs by the fp32 matrix pipe (2172 MFMAs = 174 k cycles per utterance, 85 us per 256 utterances); the same
// MACs cost 3 x 32 cycles per 16384 here, so the kernel is bound by moving and splitting x instead.
//   * workgroup = one utterance, 8 waves (two per SIMD); x is read ONCE with aligned 16-byte loads: the reference stores [F][T] contiguously
//     (src/dataset.py:52), so 16 input channels = one contiguous, 16-byte aligned slab of 16 T floats; a slab passes through LDS as
//     it is (fp32, [channel][frame]) and is split ONCE into a pixel image [frame + 1][hi 16 ch | lo 16 ch] (bf16, 64 bytes per frame,
//     swizzled, zero halo): the lane that owns frame t reads the B fragments of its three taps as two ds_read_b128 each (the layer-1
//     comment in the kernel has the pipeline);
//   * its weights (72 KB of A fragments) sit in LDS for the layer;
//   * h1 [T][32] and h2 [T][64] live in LDS channels-last as [hi: C bf16][lo: C bf16] pixels with the chunk swizzle of
//     conv3x3_mfma.h, written from the accumulator layout as 8-byte stores (4 consecutive channels of one frame per lane),
//     read as ds_read_b128 fragments at frame t + tap - 1; layers 2 and 3 keep their hi / lo A fragments in registers;
//   * frame mean and the 128 -> 1 classifier in the epilogue, as in cnn1d_fused.hip: logits[b] is the only global write.
// LDS: two fp32 slabs + two split images + layer-1 weights during layer 1 (156.5 KB at T = 321), then h1 and h2 over the same space.
// Takes the reference's storage only (x[b][f][t] contiguous, 16-byte aligned, F % 4 == 0, 3 <= T <= 347 at F = 180: the LDS budget); anything else runs
// cnn1d_fused.hip / the three-launch path (api.hip).
#include "dfa_internal.h"
#include "conv3x3_mfma.h"
#include "rng.h"

namespace dfa {
namespace c1x {
constexpr int TW = 32, NW = 8, NTH = 64 * NW, MAXT1 = 2;   // 8 waves = two per SIMD (one wave's VALU / LDS work under the other's MFMAs); layer 1: tiles w, w + 8
}

// A-fragment images: wx[m][tap][ks][part][lane] (uint4), part 0 = hi, 1 = lo; lane: co = 32 m + (lane & 31), element j <-> input
// channel 16 ks + 8 (lane >> 5) + j (zero beyond cin).  wf = BN-folded fp32 weights [cout][cin][3].
__global__ void pack_cnn1d_x3_kernel(const float* __restrict__ wf, uint4* __restrict__ wx, int cin, int cout, int nks) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int total = (cout / 32) * 3 * nks * 64;
  if (i >= total) return;
  const int lane = i & 63;
  int rest = i >> 6;
  const int ks = rest % nks; rest /= nks;
  const int tap = rest % 3, m = rest / 3;
  const int co = 32 * m + (lane & 31), hh = lane >> 5;
  bf16_t hi[8], lo[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ci = 16 * ks + 8 * hh + j;
    const float w = ci < cin ? wf[((size_t)co * cin + ci) * 3 + tap] : 0.f;
    hi[j] = float_to_bf16(w);
    lo[j] = float_to_bf16(w - bf16_to_float(hi[j]));
  }
  uint4* dst = wx + ((size_t)((m * 3 + tap) * nks + ks) * 2) * 64 + lane;
  dst[0] = *reinterpret_cast<const uint4*>(hi);
  dst[64] = *reinterpret_cast<const uint4*>(lo);
}

// k-steps of 16 input channels; layer 1 (the only layer with cin != 32, 64) runs its slabs two per loop trip: padded to an even count
int cnn1d_x3_nks(int cin) { const int n = (cin + 15) / 16; return (cin == 32 || cin == 64) ? n : (n + 1) & ~1; }
size_t cnn1d_x3_pack_bytes(int cin, int cout) { return (size_t)(cout / 32) * 3 * cnn1d_x3_nks(cin) * 2 * 64 * 16; }
hipError_t launch_pack_cnn1d_x3(const float* wf, void* wx, int cin, int cout, hipStream_t s) {
  const int total = (cout / 32) * 3 * cnn1d_x3_nks(cin) * 64;
  hipLaunchKernelGGL(pack_cnn1d_x3_kernel, dim3((total + 255) / 256), dim3(256), 0, s, wf, (uint4*)wx, cin, cout, cnn1d_x3_nks(cin));
  return hipGetLastError();
}

struct Cnn1dX3Args {
  const float* x;              // [B][F][T] contiguous, 16-byte aligned
  const uint4 *w1, *w2, *w3;   // hi / lo A-fragment images of the three layers
  const float *b1, *b2, *b3, *cw, *cb;
  float* logits;
  int T, F, NT, nks1;          // NT = ceil(T / 32), nks1 = ceil(F / 16)
  int offA_h1, offB;           // LDS byte offsets: [0, offS) two fp32 slabs, [offS, offB) two split images, [offB, ...) layer-1 weights;
  int offS, offH2;             // after layer 1: h1 at 0, h2 at offH2
  int slab_floats;             // 16 T + 8
  long long* stamps;
};

__device__ __forceinline__ f32x16_t mma_bf16(const uint4& a, const uint4& b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// 8 floats -> hi / lo bf16 fragments (element j in bf16 position j)
__device__ __forceinline__ void split8(const float (&v)[8], uint4& hi, uint4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    h[p] = pack_bf16x2(v[2 * p], v[2 * p + 1]);
    l[p] = pack_bf16x2(v[2 * p] - __uint_as_float(h[p] << 16), v[2 * p + 1] - __uint_as_float(h[p] & 0xffff0000u));
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ uint4 and4(unsigned m, const uint4& v) { return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m); }   // (a select here became an exec branch)

// one 32-frame tile of a layer whose input lives in LDS as split pixels: acc = sum over (tap, ks) of the three split products.
// PB = pixel bytes (4 * CIN), NKS = CIN / 16; lane reads pixel slot t + tap, chunks 2 ks + h (hi) and CIN / 8 + 2 ks + h (lo).
// The fragment reads run PD steps ahead of their MFMAs through a rotating register queue and `side(i)` -- a slice of the PREVIOUS
// tile's epilogue -- is issued in the shadow of step i's three MFMAs; sched_barrier pins that order (left alone, hipcc puts each
// ds_read_b128 directly in front of its MFMA and the whole epilogue between two tiles: stamps showed layers 2 / 3 at 3.6x / 2.1x
// their matrix-pipe time).
template <int CIN, typename Side>
__device__ __forceinline__ void split_gemm(const uint4 (&wh)[3 * (CIN / 16)], const uint4 (&wl)[3 * (CIN / 16)], const char* img, int t0,
                                           int col, int h, f32x16_t& acc, Side side) {
  constexpr int PB = 4 * CIN, NKS = CIN / 16, NS = 3 * NKS, PD = CIN >= 64 ? 2 : 3;   // (layer 3 holds 96 weight registers: a shorter queue)
  const char* px[3];
  int sw[3];
#pragma unroll
  for (int tap = 0; tap < 3; ++tap) {
    const int slot = t0 + col + tap;
    sw[tap] = lds_swz<PB>(slot);
    px[tap] = img + slot * PB;
  }
  uint4 qh[PD], ql[PD];
  auto rd = [&](int i, uint4& xh, uint4& xl) {
    const int tap = i / NKS, ks = i % NKS;
    xh = *(const uint4*)(px[tap] + (((2 * ks + h) ^ sw[tap]) << 4));
    xl = *(const uint4*)(px[tap] + (((CIN / 8 + 2 * ks + h) ^ sw[tap]) << 4));
  };
#pragma unroll
  for (int i = 0; i < PD; ++i) rd(i, qh[i], ql[i]);
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    const uint4 xh = qh[i % PD], xl = ql[i % PD];
    if (i + PD < NS) rd(i + PD, qh[i % PD], ql[i % PD]);
    acc = mma_bf16(wh[i], xh, acc);
    acc = mma_bf16(wl[i], xh, acc);
    acc = mma_bf16(wh[i], xl, acc);
    side(i);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// bias + ReLU + hi / lo split of one accumulator -> the split image of the next layer (COUT channels per pixel); frames >= T are
// stored as zeros (the next layer's right-hand padding); lane = frame, registers 4 g .. 4 g + 3 = channels co0 + 8 g + 4 h + (0..3)
template <int COUT>
__device__ __forceinline__ void store_split_g(const f32x16_t& acc, const float* bias, int co0, char* img, int t, int T, int h, int g) {
  constexpr int PB = 4 * COUT;
  const int slot = t + 1, sw = lds_swz<PB>(slot);
  char* px = img + slot * PB;
  const int co = co0 + 8 * g + 4 * h;
  const float4 bv = *(const float4*)(bias + co);
  float v[4] = {fmaxf(acc[4 * g] + bv.x, 0.f), fmaxf(acc[4 * g + 1] + bv.y, 0.f), fmaxf(acc[4 * g + 2] + bv.z, 0.f), fmaxf(acc[4 * g + 3] + bv.w, 0.f)};
  if (t >= T) v[0] = v[1] = v[2] = v[3] = 0.f;
  const unsigned h0 = pack_bf16x2(v[0], v[1]), h1 = pack_bf16x2(v[2], v[3]);
  const unsigned l0 = pack_bf16x2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
  const unsigned l1 = pack_bf16x2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
  const int c = co >> 3;
  *(uint2*)(px + ((c ^ sw) << 4) + 8 * h) = make_uint2(h0, h1);
  *(uint2*)(px + (((COUT / 8 + c) ^ sw) << 4) + 8 * h) = make_uint2(l0, l1);
}
template <int COUT>
__device__ __forceinline__ void store_split(const f32x16_t& acc, const float* bias, int co0, char* img, int t, int T, int h) {
#pragma unroll
  for (int g = 0; g < 4; ++g) store_split_g<COUT>(acc, bias, co0, img, t, T, h, g);
}

__global__ __launch_bounds__(512) void cnn1d_fused_x3_kernel(Cnn1dX3Args a) {
  using namespace c1x;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, h = lane >> 5;
  const int b = blockIdx.x;
  const int T = a.T, NT = a.NT;
  const int nslots = 32 * NT + 2;
  float* const slab0 = (float*)smem;                      // two slabs of slab_floats floats: [4 pad][16 x T][4 pad]
  char* const h1S = smem;                                 // region A again, after layer 1: [nslots][128 B]
  char* const w1S = smem + a.offB;                        // region B: layer-1 A fragments [3][nks1][2][64] x 16 B ...
  char* const h2S = smem + a.offH2;                       // h2 [nslots][256 B]: behind h1, over the (then dead) slab / weight regions
  float* const red = (float*)(smem + a.offH2 + nslots * 256);
  const bool stamp = a.stamps != nullptr && tid == 0 && b < 128;
  if (stamp) { a.stamps[8 * b] = __builtin_amdgcn_s_memtime(); a.stamps[8 * b + 5] = __builtin_amdgcn_s_memrealtime(); }

  // ------------------------------------------------------------------------------------------------ layer 1: F -> 32
  // A slab (16 channels x T frames, fp32, contiguous) goes global -> registers -> LDS as it is (F buffers), is then SPLIT ONCE
  // into a pixel image S[slot = frame + 1][hi: 16 ch bf16 | lo: 16 ch bf16] (64 bytes per frame, the chunk swizzle of the
  // h1 / h2 images; halo slots and the slots beyond T stay zero), and the tiles read their three taps from that image as two
  // ds_read_b128 each.  The first version split x[c][t-1..t+1] per lane and tap -- every element three times, plus six masks per
  // tile -- and layer 1 was bound by that vector work (51 k of the kernel's 86 k cycles for 13.8 k cycles of matrix-pipe time).
  // Pipeline per trip s (one barrier): compute slab s from S[s & 1] | split slab s + 1: F[(s+1) & 1] -> S[(s+1) & 1] | park slab
  // s + 2 (registers) in F[s & 1] | request slab s + 4.
  const int nks1 = a.nks1;
  f32x16_t acc1[MAXT1];
  {
    const float4* xg = (const float4*)(a.x + (size_t)b * a.F * T);
    const int SL = a.slab_floats;
    char* const S0 = smem + a.offS;                          // two split images of nslots x 64 bytes
    const int SB = nslots * 64;
    constexpr int NLD = 3;                                   // 16 T / 4 float4 per slab <= 1536 = 3 x 512
    float4 xrA[NLD], xrB[NLD];                                // even / odd slabs in flight
    const int nreal = (a.F + 15) / 16;                        // slabs that exist (nks1 may be one more: a zero slab)
    auto slab_n4 = [&](int s) { return max(0, min(16, a.F - 16 * s)) * T / 4; };
    auto slab_load = [&](int s, float4 (&xr)[NLD]) {          // unconditional, clamped index (a conditional load is an exec branch
      const int sc = min(s, nreal - 1), n4 = slab_n4(sc);    //  and hipcc's vmcnt bookkeeping across a branch is conservative)
      const float4* src = xg + (size_t)4 * sc * T;
#pragma unroll
      for (int k = 0; k < NLD; ++k) xr[k] = src[min(k * NTH + tid, n4 - 1)];
    };
    auto slab_park = [&](int s, const float4 (&xr)[NLD]) {    // registers -> F[s & 1]; channels a short / padded slab lacks become zeros
      float* dst = slab0 + (s & 1) * SL + 4;
      const int n4 = slab_n4(s);
#pragma unroll
      for (int k = 0; k < NLD; ++k) {
        const int i = k * NTH + tid;
        const unsigned m = i < n4 ? 0xffffffffu : 0u;
        const float4 v = xr[k];
        if (i < 4 * T)
          *(uint4*)(dst + 4 * i) = make_uint4(__float_as_uint(v.x) & m, __float_as_uint(v.y) & m, __float_as_uint(v.z) & m, __float_as_uint(v.w) & m);
      }
    };
    auto slab_split = [&](int s) {                            // F[s & 1] -> S[s & 1]: item = (frame t, channel octet g), lanes run along t
      const float* fb = slab0 + (s & 1) * SL + 4;
      char* sb = S0 + (s & 1) * SB;
      // 2 T <= 768 items: one per thread, the remaining 2 T - 512 go to the waves that own ONE frame tile (waves NT - 8 ...: the
      // first NT - 8 waves carry two tiles per trip and would otherwise also carry two items -- every trip ends in a barrier)
      const int t2 = tid - 64 * max(0, NT - NW);
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int it = pass == 0 ? tid : (t2 >= 0 ? NTH + t2 : 2 * T);
        if (it < 2 * T) {
          const int g = it >= T ? 1 : 0, t = it - g * T;
          float v[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) v[c] = fb[(8 * g + c) * T + t];
          uint4 hi, lo;
          split8(v, hi, lo);
          const int slot = t + 1, sw = lds_swz<64>(slot);
          *(uint4*)(sb + slot * 64 + ((g ^ sw) << 4)) = hi;
          *(uint4*)(sb + slot * 64 + (((2 + g) ^ sw) << 4)) = lo;
        }
      }
    };
    slab_load(0, xrA);
    slab_load(1, xrB);
    // layer-1 A fragments -> LDS (contiguous copy); the never-written slots of both split images (0 and T + 1 ...) -> zero
    {
      const int n = 3 * nks1 * 2 * 64;
      for (int i = tid; i < n; i += NTH) *(uint4*)(w1S + (size_t)i * 16) = a.w1[i];
      const int nz = (nslots - T) * 4;                        // 16-byte chunks of the zero slots, per image
      for (int i = tid; i < 2 * nz; i += NTH) {
        const int img = i >= nz, q = i - img * nz, zs = q >> 2;
        const int slot = zs == 0 ? 0 : T + zs;
        *(uint4*)(S0 + img * SB + slot * 64 + (q & 3) * 16) = make_uint4(0u, 0u, 0u, 0u);
      }
    }
    slab_park(0, xrA);
    slab_load(2, xrA);
    __syncthreads();
    slab_split(0);
    slab_park(1, xrB);
    slab_load(3, xrB);
    __syncthreads();

    const int nmine = (NT - wave + NW - 1) / NW;
#pragma unroll
    for (int j = 0; j < MAXT1; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
    auto trip = [&](int s, float4 (&xr)[NLD]) {               // xr: the register set of this trip's parity (holds slab s + 2)
      const char* sb = S0 + (s & 1) * SB;
      uint4 wh[3], wl[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        wh[k] = *(const uint4*)(w1S + ((size_t)((k * nks1 + s) * 2) * 64 + lane) * 16);
        wl[k] = *(const uint4*)(w1S + ((size_t)((k * nks1 + s) * 2 + 1) * 64 + lane) * 16);
      }
      uint4 xh[MAXT1][3], xl[MAXT1][3];
#pragma unroll
      for (int j = 0; j < MAXT1; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int slot = min(TW * (wave + NW * j) + col + k, nslots - 1);      // (tiles this wave does not have: clamped, unused)
          const int sw = lds_swz<64>(slot);
          xh[j][k] = *(const uint4*)(sb + slot * 64 + ((h ^ sw) << 4));
          xl[j][k] = *(const uint4*)(sb + slot * 64 + (((2 + h) ^ sw) << 4));
        }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < MAXT1; ++j)
        if (j < nmine) {
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            acc1[j] = mma_bf16(wh[k], xh[j][k], acc1[j]);
            acc1[j] = mma_bf16(wl[k], xh[j][k], acc1[j]);
            acc1[j] = mma_bf16(wh[k], xl[j][k], acc1[j]);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
      slab_split(s + 1);                                      // (past the last slab: a stale buffer into an image nobody reads)
      slab_park(s + 2, xr);
      slab_load(s + 4, xr);                                   // (clamped: past the end it re-reads the last slab, never parked as data)
      __syncthreads();
    };
    for (int s = 0; s < nks1; s += 2) {                       // nks1 is even
      trip(s, xrA);
      trip(s + 1, xrB);
    }
  }
  if (stamp) a.stamps[8 * b + 1] = __builtin_amdgcn_s_memtime();
  // (the barrier that closed the loop: every wave is done with the slabs and the layer-1 weights)
  {
    // h1 halo: slot 0 (frame -1); frames >= T are written as zeros by the epilogue below, slot 32 NT + 1 here
    if (tid < 16) *(uint4*)(h1S + (tid < 8 ? 0 : (nslots - 1) * 128) + (tid & 7) * 16) = make_uint4(0u, 0u, 0u, 0u);
    const int nmine = (NT - wave + NW - 1) / NW;
#pragma unroll
    for (int j = 0; j < MAXT1; ++j)
      if (j < nmine) store_split<32>(acc1[j], a.b1, 0, h1S, TW * (wave + NW * j) + col, T, h);
  }
  // hi / lo weight fragments of layer 2 (this wave's 32 output channels): requested in front of the barrier
  uint4 w2h[6], w2l[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    w2h[i] = a.w2[((size_t)((wave & 1) * 6 + i) * 2) * 64 + lane];
    w2l[i] = a.w2[((size_t)((wave & 1) * 6 + i) * 2 + 1) * 64 + lane];
  }
  __syncthreads();

  // ------------------------------------------------------------------------------------------------ layer 2: 32 -> 64
  {
    const int m = wave & 1, par = wave >> 1;                 // 32 of the 64 channels; tiles par, par + 4, par + 8
    constexpr int ST = NW / 2;
    if (tid < 32) *(uint4*)(h2S + (tid < 16 ? 0 : (nslots - 1) * 256) + (tid & 15) * 16) = make_uint4(0u, 0u, 0u, 0u);
    // tile i's bias + ReLU + split + store rides on tile i + 1's steps (one 4-channel group per step)
    f32x16_t accA, accB;
    int tile = par;
    if (tile < NT) {
      split_gemm<32>(w2h, w2l, h1S, TW * tile, col, h, accA, [](int) {});
      for (tile += ST; tile + ST < NT; tile += 2 * ST) {
        split_gemm<32>(w2h, w2l, h1S, TW * tile, col, h, accB,
                       [&](int i) { if (i >= 1 && i < 5) store_split_g<64>(accA, a.b2, 32 * m, h2S, TW * (tile - ST) + col, T, h, i - 1); });
        split_gemm<32>(w2h, w2l, h1S, TW * (tile + ST), col, h, accA,
                       [&](int i) { if (i >= 1 && i < 5) store_split_g<64>(accB, a.b2, 32 * m, h2S, TW * tile + col, T, h, i - 1); });
      }
      if (tile < NT) {
        split_gemm<32>(w2h, w2l, h1S, TW * tile, col, h, accB,
                       [&](int i) { if (i >= 1 && i < 5) store_split_g<64>(accA, a.b2, 32 * m, h2S, TW * (tile - ST) + col, T, h, i - 1); });
        store_split<64>(accB, a.b2, 32 * m, h2S, TW * tile + col, T, h);
      } else {
        store_split<64>(accA, a.b2, 32 * m, h2S, TW * (tile - ST) + col, T, h);
      }
    }
  }
  // layer 3's fragments (32 of the 128 channels per wave; two waves share a channel tile and split its frame tiles)
  uint4 w3h[12], w3l[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    w3h[i] = a.w3[((size_t)((wave & 3) * 12 + i) * 2) * 64 + lane];
    w3l[i] = a.w3[((size_t)((wave & 3) * 12 + i) * 2 + 1) * 64 + lane];
  }
  __syncthreads();
  if (stamp) a.stamps[8 * b + 2] = __builtin_amdgcn_s_memtime();

  // ------------------------------------------------------------------------------------------------ layer 3: 64 -> 128, frame mean, classifier
  {
    const int m = wave & 3, par = wave >> 2;                 // tiles par, par + 2, ...
    float bias[16], sum[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      bias[r] = a.b3[32 * m + (r & 3) + 8 * (r >> 2) + 4 * h];
      sum[r] = 0.f;
    }
    auto add_regs = [&](const f32x16_t& v, int tile, int r0) {       // two accumulator registers per step
      const bool inside = TW * tile + col < T;
#pragma unroll
      for (int r = r0; r < r0 + 2; ++r) sum[r] += inside ? fmaxf(v[r] + bias[r], 0.f) : 0.f;
    };
    f32x16_t accA, accB;
    int tile = par;
    if (tile < NT) {
      split_gemm<64>(w3h, w3l, h2S, TW * tile, col, h, accA, [](int) {});
      for (tile += 2; tile + 2 < NT; tile += 4) {
        split_gemm<64>(w3h, w3l, h2S, TW * tile, col, h, accB, [&](int i) { if (i >= 2 && i < 10) add_regs(accA, tile - 2, 2 * (i - 2)); });
        split_gemm<64>(w3h, w3l, h2S, TW * (tile + 2), col, h, accA, [&](int i) { if (i >= 2 && i < 10) add_regs(accB, tile, 2 * (i - 2)); });
      }
      if (tile < NT) {
        split_gemm<64>(w3h, w3l, h2S, TW * tile, col, h, accB, [&](int i) { if (i >= 2 && i < 10) add_regs(accA, tile - 2, 2 * (i - 2)); });
#pragma unroll
        for (int r = 0; r < 16; r += 2) add_regs(accB, tile, r);
      } else {
#pragma unroll
        for (int r = 0; r < 16; r += 2) add_regs(accA, tile - 2, r);
      }
    }
    float part = 0.f;
    const float inv_t = 1.0f / (float)T;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s = sum[r];
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      part = fmaf(s * inv_t, a.cw[32 * m + (r & 3) + 8 * (r >> 2) + 4 * h], part);
    }
    part += __shfl_xor(part, 32, 64);
    if (lane == 0) red[wave] = part;
  }
  __syncthreads();
  if (stamp) { a.stamps[8 * b + 3] = __builtin_amdgcn_s_memtime(); a.stamps[8 * b + 6] = __builtin_amdgcn_s_memrealtime(); }
  if (tid == 0) a.logits[b] = (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]))) + a.cb[0];
}

static void cnn1d_x3_layout(int T, int F, int* offB, int* total, int* slab_floats, int* offS = nullptr, int* offH2 = nullptr) {
  const int NT = (T + 31) / 32, nslots = 32 * NT + 2, nks1 = cnn1d_x3_nks(F);
  const int SL = 16 * T + 8;
  const int oS = (2 * SL * 4 + 255) / 256 * 256;                       // two fp32 slabs
  const int oB = (oS + 2 * nslots * 64 + 255) / 256 * 256;             // two split images
  const int oH2 = (nslots * 128 + 255) / 256 * 256;                    // h1, then h2
  *offB = oB;
  *total = std::max(oB + 3 * nks1 * 2 * 64 * 16, oH2 + nslots * 256 + 64);
  *slab_floats = SL;
  if (offS) *offS = oS;
  if (offH2) *offH2 = oH2;
}
// x must be the contiguous [B][F][T] storage (element (b, t, f) at b F T + f T + t), 16-byte aligned
// ---- one Conv1d(k = 3, pad 1) layer of the TRAINING step on the matrix cores (src/train.py:71-76 through
// src/model_cnn1d.py:17-34): z[b][co][t] = bias[co] + sum_{ci, k} W[co][ci][k] in[b][ci][t + k - 1], channel-major fp32 in and
// out, no BatchNorm / ReLU (train mode: batch statistics come first).  It serves the three forward convolutions and -- on the
// flipped / transposed weight image of conv1d_dgrad_pack_kernel -- the two data gradients.  Layer 1 of the fused kernel above,
// generalised: a workgroup = (utterance, MT tiles of 32 output channels); 16-channel slabs of the input go through LDS as they are
// (contiguous 16 T floats, 16-byte loads, two slabs ahead), the lane that owns frame t splits x[c][t-1..t+1] of its 8 channels
// into bf16 B fragments ONCE per slab and tap and feeds them to every channel tile; this layer's A fragments sit in LDS.
// TERMS = 3 (default): every fp32 operand = hi + lo + lo2, three bf16 terms = its 24-bit mantissa exactly, six MFMAs per product
// (all term pairs of order <= 2; the dropped ones are below 2^-24 of a product): fp32-grade sums.  TERMS = 2: the bf16x3
// construction of the eval kernel (16-bit operands, three MFMAs, ~1e-5) -- fine for inference's 1e-4 bar, but in a training step
// a 1e-5 perturbation of a pre-activation flips ~1e-5 of the ReLU masks and every flip moves a gradient by a whole element:
// relative L2 error ~ sqrt(1e-5) = 3e-3 against 3e-4 for fp32 arithmetic (tools/gpu_cnn1d_x3_probe.py), so it is opt-in.
// The fp32 VALU kernel this replaces (conv1d.hip) needs 97 us per call at [256, *, 321].
struct Conv1dX3Args {
  const float* x;              // [B] x [Cin][T] contiguous per utterance, utterance stride sb floats, 16-byte aligned
  int64_t sb;
  const uint4* w;              // pack_conv1d_terms_kernel image [Cout / 32][3][nks][TERMS][64]
  const float* bias;           // [Cout]
  float* z;                    // [B][Cout][T]
  int T, Cin, Cout, NT, nks, slab_floats, offW;
  AugCfg aug;                  // AUG instantiation (training layer 1): x is read through the armed train-time augmentation (rng.h)
};

// A-fragment images with TERMS bf16 terms per weight: wx[m][tap][ks][term][lane]
__global__ void pack_conv1d_terms_kernel(const float* __restrict__ wf, uint4* __restrict__ wx, int cin, int cout, int nks, int terms) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int total = (cout / 32) * 3 * nks * 64;
  if (i >= total) return;
  const int lane = i & 63;
  int rest = i >> 6;
  const int ks = rest % nks; rest /= nks;
  const int tap = rest % 3, m = rest / 3;
  const int co = 32 * m + (lane & 31), hh = lane >> 5;
  bf16_t t0[8], t1[8], t2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ci = 16 * ks + 8 * hh + j;
    const float w = ci < cin ? wf[((size_t)co * cin + ci) * 3 + tap] : 0.f;
    t0[j] = float_to_bf16(w);
    const float r1 = w - bf16_to_float(t0[j]);
    t1[j] = float_to_bf16(r1);
    t2[j] = float_to_bf16(r1 - bf16_to_float(t1[j]));
  }
  uint4* dst = wx + ((size_t)((m * 3 + tap) * nks + ks) * terms) * 64 + lane;
  dst[0] = *reinterpret_cast<const uint4*>(t0);
  dst[64] = *reinterpret_cast<const uint4*>(t1);
  if (terms == 3) dst[128] = *reinterpret_cast<const uint4*>(t2);
}
size_t conv1d_terms_pack_bytes(int cin, int cout, int terms) { return (size_t)(cout / 32) * 3 * cnn1d_x3_nks(cin) * terms * 64 * 16; }
hipError_t launch_pack_conv1d_terms(const float* wf, void* wx, int cin, int cout, int terms, hipStream_t s) {
  const int total = (cout / 32) * 3 * cnn1d_x3_nks(cin) * 64;
  hipLaunchKernelGGL(pack_conv1d_terms_kernel, dim3((total + 255) / 256), dim3(256), 0, s, wf, (uint4*)wx, cin, cout, cnn1d_x3_nks(cin), terms);
  return hipGetLastError();
}

// The five images of a training step (forward layers 1-3, data gradients 3 -> 2 and 2 -> 1) in ONE launch, the data-gradient ones read
// straight from the layer's own weight: W'[c][o][k'] = W[o][c][2 - k'] (a Conv1d with Cin' = Cout, Cout' = Cin) -- seven launches
// (five packs + two transposes) of ~4.5 us each otherwise, in a step of 0.6 ms.
struct Conv1dPackAll {
  const float* w[5];     // the layer's weight [Cout][Cin][3] (for a flipped entry: the FORWARD layer's weight)
  uint4* dst[5];
  int cin[5], cout[5], nks[5], flip[5], begin[6];   // cin / cout of the convolution the image serves; fragment range [begin[i], begin[i+1])
  int terms;
  float* zero_bias;      // [256] zeroed here (the data gradients' bias)
};
__global__ void pack_conv1d_train_all_kernel(Conv1dPackAll a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < 256) a.zero_bias[i] = 0.f;
  if (i >= a.begin[5]) return;
  int e = 0;
#pragma unroll
  for (int q = 1; q < 5; ++q) e += (i >= a.begin[q]) ? 1 : 0;
  const int li = i - a.begin[e], cin = a.cin[e], cout = a.cout[e], nks = a.nks[e];
  const int lane = li & 63;
  int rest = li >> 6;
  const int ks = rest % nks; rest /= nks;
  const int tap = rest % 3, m = rest / 3;
  const int co = 32 * m + (lane & 31), hh = lane >> 5;
  const float* w = a.w[e];
  bf16_t t0[8], t1[8], t2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int ci = 16 * ks + 8 * hh + j;
    float v = 0.f;
    if (ci < cin) v = a.flip[e] ? w[((size_t)ci * cout + co) * 3 + (2 - tap)]      // forward weight [o = ci'][c = co'][2 - k'], its Cin = cout'
                                : w[((size_t)co * cin + ci) * 3 + tap];
    t0[j] = float_to_bf16(v);
    const float r1 = v - bf16_to_float(t0[j]);
    t1[j] = float_to_bf16(r1);
    t2[j] = float_to_bf16(r1 - bf16_to_float(t1[j]));
  }
  uint4* dst = a.dst[e] + ((size_t)((m * 3 + tap) * nks + ks) * a.terms) * 64 + lane;
  dst[0] = *reinterpret_cast<const uint4*>(t0);
  dst[64] = *reinterpret_cast<const uint4*>(t1);
  if (a.terms == 3) dst[128] = *reinterpret_cast<const uint4*>(t2);
}
// w1, w2, w3: the three Conv1d weights [32][F][3], [64][32][3], [128][64][3]; dst[0..4]: forward 1-3, data gradients 3 -> 2, 2 -> 1
hipError_t launch_pack_conv1d_train_all(const float* w1, const float* w2, const float* w3, void* const* dst, int F, int terms, float* zero_bias,
                                        hipStream_t s) {
  Conv1dPackAll a{};
  const float* w[5] = {w1, w2, w3, w3, w2};
  const int cin[5] = {F, 32, 64, 128, 64}, cout[5] = {32, 64, 128, 64, 32}, flip[5] = {0, 0, 0, 1, 1};
  a.begin[0] = 0;
  for (int i = 0; i < 5; ++i) {
    a.w[i] = w[i]; a.dst[i] = (uint4*)dst[i]; a.cin[i] = cin[i]; a.cout[i] = cout[i]; a.flip[i] = flip[i];
    a.nks[i] = cnn1d_x3_nks(cin[i]);
    a.begin[i + 1] = a.begin[i] + (cout[i] / 32) * 3 * a.nks[i] * 64;
  }
  a.terms = terms; a.zero_bias = zero_bias;
  const int total = a.begin[5] > 256 ? a.begin[5] : 256;
  hipLaunchKernelGGL(pack_conv1d_train_all_kernel, dim3((total + 255) / 256), dim3(256), 0, s, a);
  return hipGetLastError();
}

// 8 floats -> TERMS bf16 fragments (element j in bf16 position j): v = f[0] + f[1] (+ f[2]), exactly for TERMS = 3
template <int TERMS>
__device__ __forceinline__ void split8n(const float (&v)[8], uint4 (&f)[TERMS]) {
  unsigned q[TERMS][4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    float r0 = v[2 * p], r1 = v[2 * p + 1];
#pragma unroll
    for (int t = 0; t < TERMS; ++t) {
      q[t][p] = pack_bf16x2(r0, r1);
      if (t + 1 < TERMS) { r0 -= __uint_as_float(q[t][p] << 16); r1 -= __uint_as_float(q[t][p] & 0xffff0000u); }
    }
  }
#pragma unroll
  for (int t = 0; t < TERMS; ++t) f[t] = make_uint4(q[t][0], q[t][1], q[t][2], q[t][3]);
}

template <int MT, int TERMS, bool AUG = false>
__global__ __launch_bounds__(512) void conv1d_x3_kernel(Conv1dX3Args a) {
  using namespace c1x;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, h = lane >> 5;
  const int b = blockIdx.x, m0 = blockIdx.y * MT;
  const int T = a.T, NT = a.NT, nks = a.nks;
  float* const slab0 = (float*)smem;                      // two slabs of slab_floats floats: [4 pad][16 x T][4 pad]
  char* const wS = smem + a.offW;                         // this workgroup's A fragments [MT][3][nks][TERMS][64] x 16 B
  f32x16_t acc[MT][MAXT1];
  const float4* xg = (const float4*)(a.x + (size_t)b * a.sb);
  const int SL = a.slab_floats;
  constexpr int NLD = 3;                                   // 16 T / 4 float4 per slab <= 1536 = 3 x 512
  float4 xrA[NLD], xrB[NLD];
  const int nreal = (a.Cin + 15) / 16;                     // slabs that exist (nks may be one more: a zero slab)
  auto slab_n4 = [&](int s) { return max(0, min(16, a.Cin - 16 * s)) * T / 4; };
  auto slab_load = [&](int s, float4 (&xr)[NLD]) {         // unconditional, clamped (see the fused kernel)
    const int sc = min(s, nreal - 1), n4 = slab_n4(sc);
    if constexpr (AUG) {
      // element (channel = feature dim f, frame t) of the augmented batch: the source frame is rolled, so the slab is gathered
      // element by element (4 dword loads per float4 slot) and finished by aug_apply -- the value dfa_augment_batch would write
      const float* xf = a.x + (size_t)b * a.sb;
#pragma unroll
      for (int k = 0; k < NLD; ++k) {
        const int i = min(k * NTH + tid, n4 - 1);
        float e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int flat = 4 * i + u, c = flat / T, t = flat - c * T, f = 16 * sc + c;
          e[u] = aug_apply(a.aug, xf[(size_t)f * T + aug_src_t(a.aug, t)], b, t, f);
        }
        xr[k] = make_float4(e[0], e[1], e[2], e[3]);
      }
    } else {
      const float4* src = xg + (size_t)4 * sc * T;
#pragma unroll
      for (int k = 0; k < NLD; ++k) xr[k] = src[min(k * NTH + tid, n4 - 1)];
    }
  };
  auto slab_store = [&](int s, const float4 (&xr)[NLD]) {
    float* dst = slab0 + (s & 1) * SL + 4;
    const int n4 = slab_n4(s);
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int i = k * NTH + tid;
      const unsigned m = i < n4 ? 0xffffffffu : 0u;
      const float4 v = xr[k];
      if (i < 4 * T)
        *(uint4*)(dst + 4 * i) = make_uint4(__float_as_uint(v.x) & m, __float_as_uint(v.y) & m, __float_as_uint(v.z) & m, __float_as_uint(v.w) & m);
    }
  };
  slab_load(0, xrA);
  slab_load(1, xrB);
  {
    const int n = MT * 3 * nks * TERMS * 64;
    const uint4* wsrc = a.w + (size_t)m0 * 3 * nks * TERMS * 64;
    for (int i = tid; i < n; i += NTH) *(uint4*)(wS + (size_t)i * 16) = wsrc[i];
    if (tid < 16) {
      const int sbuf = tid >> 3, e = tid & 7;
      slab0[sbuf * SL + (e < 4 ? e : 16 * T + e)] = 0.f;
    }
  }
  slab_store(0, xrA);
  __syncthreads();

  const int nmine = (NT - wave + NW - 1) / NW;
  unsigned tin[MAXT1][3];                                  // all-ones where tap k of this lane's frame exists, else 0
  int tl[MAXT1];
#pragma unroll
  for (int j = 0; j < MAXT1; ++j) {
    const int t = TW * (wave + NW * j) + col;
    tl[j] = t;
#pragma unroll
    for (int k = 0; k < 3; ++k) tin[j][k] = ((j < nmine) && t < T && t - 1 + k >= 0 && t - 1 + k < T) ? 0xffffffffu : 0u;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
  }
  // one slab = one k-step of 16 channels: trip s computes slab s from buffer s & 1, requests slab s + 2 and parks slab s + 1
  // (requested a trip ago) in the other buffer; one barrier per trip.  Inside a trip the TAP is the outer loop: its MT x TERMS
  // weight fragments are read from LDS once and serve both of the wave's frame tiles.
  auto trip = [&](int s, float4 (&xr_next)[NLD], float4 (&xr_far)[NLD], bool load) {
    if (load) slab_load(s + 2, xr_far);
    const float* sl = slab0 + (s & 1) * SL + 4 + 8 * h * T - 1;       // this lane half's 8 channels, frame index - 1
    float v[MAXT1][3][8];
#pragma unroll
    for (int j = 0; j < MAXT1; ++j)
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int k = 0; k < 3; ++k) v[j][k][c] = sl[c * T + tl[j] + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      uint4 w[MT][TERMS];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int p = 0; p < TERMS; ++p) w[m][p] = *(const uint4*)(wS + ((size_t)((((m * 3 + k) * nks + s) * TERMS + p) * 64 + lane)) * 16);
#pragma unroll
      for (int j = 0; j < MAXT1; ++j) {
        uint4 xs[TERMS];
        split8n<TERMS>(v[j][k], xs);
#pragma unroll
        for (int p = 0; p < TERMS; ++p) xs[p] = and4(tin[j][k], xs[p]);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          if constexpr (TERMS == 3) {                      // smallest terms first
            acc[m][j] = mma_bf16(w[m][2], xs[0], acc[m][j]);
            acc[m][j] = mma_bf16(w[m][0], xs[2], acc[m][j]);
            acc[m][j] = mma_bf16(w[m][1], xs[1], acc[m][j]);
          }
          acc[m][j] = mma_bf16(w[m][1], xs[0], acc[m][j]);
          acc[m][j] = mma_bf16(w[m][0], xs[1], acc[m][j]);
          acc[m][j] = mma_bf16(w[m][0], xs[0], acc[m][j]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    slab_store(s + 1, xr_next);
    __syncthreads();
  };
  int s = 0;
  for (; s + 2 < nks; s += 2) {                              // nks is even
    trip(s, xrB, xrA, true);
    trip(s + 1, xrA, xrB, true);
  }
  trip(s, xrB, xrA, false);
  trip(s + 1, xrA, xrB, false);

  // ---- + bias, channel-major store: register r = channel (r&3) + 8 (r>>2) + 4 h of frame t = lane: 32 lanes write 128 contiguous bytes
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int cb = 32 * (m0 + m) + 4 * h;
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = a.bias[cb + (r & 3) + 8 * (r >> 2)];
#pragma unroll
    for (int j = 0; j < MAXT1; ++j) {
      const int t = tl[j];
      if (j < nmine && t < T) {
        float* zr = a.z + ((size_t)b * a.Cout + cb) * T + t;
#pragma unroll
        for (int r = 0; r < 16; ++r) zr[(size_t)((r & 3) + 8 * (r >> 2)) * T] = acc[m][j][r] + bv[r];
      }
    }
  }
}

static void conv1d_x3_layout(int T, int cin, int mt, int terms, int* offW, int* total, int* slab_floats) {
  const int sl = 16 * T + 8;
  *slab_floats = sl;
  *offW = (2 * sl * 4 + 255) & ~255;
  *total = *offW + mt * 3 * cnn1d_x3_nks(cin) * terms * 64 * 16;
}
// channel tiles per workgroup: two when the layer has them and their fragments fit beside the slabs; mode 2 (diagnostic) forces one
static int conv1d_x3_mt(int T, int cin, int cout, int terms, int mode) {
  if (mode == 2 || cout % 64 != 0) return 1;
  int offW, total, sl;
  conv1d_x3_layout(T, cin, 2, terms, &offW, &total, &sl);
  return total <= 160 * 1024 ? 2 : 1;
}

bool conv1d_x3_supports(const float* x, int64_t sb, int64_t sc, int64_t st, const float* z, int T, int Cin, int Cout, int terms) {
  if (T < 3 || T > 384 || (T + 31) / 32 > c1x::NW * c1x::MAXT1 || Cin < 1 || (Cin & 3) || Cout < 32 || (Cout & 31)) return false;
  if (st != 1 || sc != T || (sb & 3) || ((uintptr_t)x & 15) || ((uintptr_t)z & 3)) return false;
  if ((cnn1d_x3_nks(Cin) & 1) || (terms != 2 && terms != 3)) return false;
  int offW, total, sl;
  conv1d_x3_layout(T, Cin, 1, terms, &offW, &total, &sl);
  return total <= 160 * 1024;
}

hipError_t launch_conv1d_x3(const float* x, int64_t sb, const void* wx, const float* bias, float* z, int B, int Cin, int Cout, int T,
                            int terms, hipStream_t s, int mode, const AugCfg* aug) {
  Conv1dX3Args a{};
  if (aug && aug->on) a.aug = *aug;
  a.x = x; a.sb = sb; a.w = (const uint4*)wx; a.bias = bias; a.z = z; a.T = T; a.Cin = Cin; a.Cout = Cout;
  a.NT = (T + 31) / 32; a.nks = cnn1d_x3_nks(Cin);
  const int mt = conv1d_x3_mt(T, Cin, Cout, terms, mode);
  int total;
  conv1d_x3_layout(T, Cin, mt, terms, &a.offW, &total, &a.slab_floats);
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(B, Cout / (32 * mt)), dim3(c1x::NTH), total, s, a);
    return hipGetLastError();
  };
  if (a.aug.on) {                 // layer 1 only (one channel tile)
    if (mt != 1) return hipErrorInvalidValue;
    return terms == 3 ? go(conv1d_x3_kernel<1, 3, true>) : go(conv1d_x3_kernel<1, 2, true>);
  }
  if (terms == 3) return mt == 2 ? go(conv1d_x3_kernel<2, 3>) : go(conv1d_x3_kernel<1, 3>);
  return mt == 2 ? go(conv1d_x3_kernel<2, 2>) : go(conv1d_x3_kernel<1, 2>);
}

bool cnn1d_fused_x3_supports(const void* x, int64_t sb, int64_t st, int64_t sf, int T, int F) {
  // (a slab of 16 channels = 4 T float4 must fit the 3 x 512 loads of a trip: T <= 384)
  if (T < 3 || T > 384 || (T + 31) / 32 > c1x::NW * c1x::MAXT1 || F < 1 || (F & 3)) return false;
  if (st != 1 || sf != T || sb != (int64_t)F * T || ((uintptr_t)x & 15)) return false;
  int offB, total, sl;
  cnn1d_x3_layout(T, F, &offB, &total, &sl);
  return total <= 160 * 1024;
}

hipError_t launch_cnn1d_fused_x3(const float* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3,
                                 const float* b3, const float* cw, const float* cb, float* logits, int B, int T, int F, hipStream_t s,
                                 long long* stamps) {
  Cnn1dX3Args a{};
  a.x = x; a.w1 = (const uint4*)w1; a.w2 = (const uint4*)w2; a.w3 = (const uint4*)w3; a.b1 = b1; a.b2 = b2; a.b3 = b3; a.cw = cw; a.cb = cb;
  a.logits = logits; a.T = T; a.F = F; a.NT = (T + 31) / 32; a.nks1 = cnn1d_x3_nks(F); a.stamps = stamps;
  int total;
  cnn1d_x3_layout(T, F, &a.offB, &total, &a.slab_floats, &a.offS, &a.offH2);
  hipError_t e = hipFuncSetAttribute((const void*)cnn1d_fused_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(cnn1d_fused_x3_kernel, dim3(B), dim3(c1x::NTH), total, s, a);
  return hipGetLastError();
}

}  // namespace dfa
