// wgrad_mfma.hip -- weight gradient of the 3x3 / pad 1 convolutions on the matrix cores (autograd of
// src/model.py:21,27 inside loss.backward(), src/train.py:75):
//     dW[co][ci][dy][dx] = sum_{b,t,f} dz[b][t][f][co] * a[b][t+dy-1][f+dx-1][ci],     db[co] = sum dz[b][t][f][co]
// Nine GEMMs  [COUT x pixels] . [pixels x CIN]  that share the A operand (dz) and differ only by the tap shift of
// the B operand (a).  The reduction dimension is the pixel index, so with channels-last activations one fp32
// v_mfma_f32_32x32x2_f32 takes its operands straight from row-major LDS tiles (lane = channel, k = pixel parity):
// no transposed copies are needed.  The arithmetic is fp32 in both precision modes (bf16 activations are widened
// while staging).  Each workgroup walks its share of (utterance, row, 64-column segment) work items keeping all of
// its 32x32 accumulator tiles in registers, and writes ONE partial dW at the end; a fixed-order fp64 reduction over
// the workgroups follows (deterministic, no atomics).
#include "dfa_internal.h"

namespace dfa {

template <typename T>
__device__ __forceinline__ void widen8(const T* p, float* v);
template <>
__device__ __forceinline__ void widen8<float>(const float* p, float* v) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void widen8<bf16_t>(const bf16_t* p, float* v) {
  const uint4 q = *reinterpret_cast<const uint4*>(p);
  const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(u[e] << 16); v[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
}

constexpr int WG_SEG = 64;  // pixels (columns of one row) per work item

// tiles: (co slice cs, ci slice is, tap).  CS = COUT/32 must be 4 or 2.
//   CS == 4: wave w owns cs = w, all IS*9 (is, tap) tiles.
//   CS == 2: wave w owns cs = w & 1 and taps [0,5) (w < 2) or [5,9) (w >= 2), all is.
template <typename T, int CIN, int COUT>
__global__ __launch_bounds__(256, 1) void wgrad3x3_mfma_kernel(const T* __restrict__ dz, const T* __restrict__ a,
                                                               float* __restrict__ partial, int B, int H, int W,
                                                               int dzs_c, int as_c) {   // channels per pixel in memory
  constexpr int CS = COUT / 32, IS = CIN / 32;
  static_assert(CS == 4 || CS == 2, "COUT must be 64 or 128");
  constexpr int NTAP = (CS == 4) ? 9 : 5;       // max taps per wave
  constexpr int NTILE = IS * NTAP;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dzs = (float*)smem;                      // [WG_SEG][COUT]
  float* as = dzs + WG_SEG * COUT;                // [3][WG_SEG + 2][CIN]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int cs = (CS == 4) ? wave : (wave & 1);
  const int tap0 = (CS == 4) ? 0 : ((wave >> 1) ? 5 : 0);
  const int ntap = (CS == 4) ? 9 : ((wave >> 1) ? 4 : 5);

  f32x16_t acc[NTILE];
#pragma unroll
  for (int q = 0; q < NTILE; ++q)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
  float dbsum = 0.f;

  const int nseg = (W + WG_SEG - 1) / WG_SEG;
  const long nitems = (long)B * H * nseg;
  for (long item = blockIdx.x; item < nitems; item += gridDim.x) {
    const int seg = (int)(item % nseg);
    const long bt = item / nseg;
    const int t = (int)(bt % H), b = (int)(bt / H);
    const int f0 = seg * WG_SEG;
    __syncthreads();  // previous item's MFMAs are done with the tiles
    // stage dz row segment (zero beyond W) and the 3 x (64+2) halo tile of a, widened to fp32
    for (int e = tid; e < WG_SEG * (COUT / 8); e += 256) {
      const int p = e / (COUT / 8), cg = e % (COUT / 8);
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (f0 + p < W) widen8<T>(dz + ((((size_t)b * H + t) * W + f0 + p) * dzs_c + cg * 8), v);
      float4* d = reinterpret_cast<float4*>(dzs + p * COUT + cg * 8);
      d[0] = make_float4(v[0], v[1], v[2], v[3]);
      d[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    for (int e = tid; e < 3 * (WG_SEG + 2) * (CIN / 8); e += 256) {
      const int cg = e % (CIN / 8);
      const int sl = (e / (CIN / 8)) % (WG_SEG + 2), row = e / ((CIN / 8) * (WG_SEG + 2));
      const int tt = t + row - 1, ff = f0 - 1 + sl;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (tt >= 0 && tt < H && ff >= 0 && ff < W) widen8<T>(a + ((((size_t)b * H + tt) * W + ff) * as_c + cg * 8), v);
      float4* d = reinterpret_cast<float4*>(as + (row * (WG_SEG + 2) + sl) * CIN + cg * 8);
      d[0] = make_float4(v[0], v[1], v[2], v[3]);
      d[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    __syncthreads();
#pragma unroll 2
    for (int kk = 0; kk < WG_SEG / 2; ++kk) {
      const int p = 2 * kk + h;
      const float av = dzs[p * COUT + cs * 32 + r];
      dbsum += av;
#pragma unroll
      for (int tp = 0; tp < NTAP; ++tp) {
        if (tp < ntap) {
          const int tap = tap0 + tp;
          const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
          for (int is = 0; is < IS; ++is) {
            const float bv = as[(dy * (WG_SEG + 2) + p + dx) * CIN + is * 32 + r];
            acc[tp * IS + is] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tp * IS + is], 0, 0, 0);
          }
        }
      }
    }
  }
  // partial[blockIdx.x][COUT][CIN][9] (+ [COUT] bias sums at the end)
  float* out = partial + (size_t)blockIdx.x * ((size_t)COUT * CIN * 9 + COUT);
#pragma unroll
  for (int tp = 0; tp < NTAP; ++tp) {
    if (tp < ntap) {
      const int tap = tap0 + tp;
#pragma unroll
      for (int is = 0; is < IS; ++is)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int co = cs * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, ci = is * 32 + r;
          out[((size_t)co * CIN + ci) * 9 + tap] = acc[tp * IS + is][i];
        }
    }
  }
  dbsum += __shfl_xor(dbsum, 32, 64);
  if (h == 0 && tap0 == 0) out[(size_t)COUT * CIN * 9 + cs * 32 + r] = dbsum;
}

// ------------------------------------------------------------------------------------------------ bf16 kernel
// Same decomposition on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate).  The MFMA wants 8 CONSECUTIVE k (= pixels)
// per lane for one channel, i.e. a column of the channels-last LDS tile: ds_read_b64_tr_b16 delivers exactly that (a
// 4-pixel x 16-channel block transposed per 16-lane group), and because a pixel is a ROW of the tile the 3x3 tap shift
// is a plain row offset -- no unaligned accesses.  Pixel rows are padded to a stride == 64 (mod 128) bytes so the 4
// rows x 64 bytes a half-wave touches fall on disjoint bank windows (conflict-free).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

constexpr int wg_stride(int c) { return (c * 2 % 128 == 64) ? c * 2 : c * 2 + 64; }

// ---- bf16 kernel, second version (kept as the reference the tests compare v3 against).  The first one (removed) gave
// every wave ONE 32-channel dz slice: each transposed activation
// fragment (2 x ds_read_b64_tr_b16) fed a single MFMA, LDS reads were waited for one by one at one wave per SIMD, and the
// tiles were staged synchronously: 0.29 PFLOP/s.  Here a wave owns 18 accumulator tiles = 9 taps x 2 dz slices (288
// AGPRs at one wave per SIMD): a k-step of 16 pixels reads 2 dz + 9 activation fragments for 18 MFMAs (0.6 fragment
// reads per MFMA, all issued before the first MFMA needs them), and the next work item's tiles are fetched into
// registers while the current one is computed (double-buffered LDS, one barrier per item).
//   KSPLIT = false (64 -> 128): the four waves are (ci slice, dz-slice pair) on the same pixels.
//   KSPLIT = true  (32 -> 64):  18 tiles are the whole gradient: the four waves take different k-steps (pixels) of the
//                               item and each writes its own partial record.
template <int CIN, int COUT, int SEG, bool KSPLIT>
__global__ __launch_bounds__(256, 1) void wgrad3x3_bf16_v2_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ a,
                                                                  float* __restrict__ partial, int B, int H, int W,
                                                                  int dzs_c, int as_c) {
  constexpr int CS = COUT / 32, IS = CIN / 32;
  static_assert(KSPLIT ? (IS == 1 && CS == 2) : (IS * CS == 8 && IS == 2), "tile split covers (32,64) and (64,128)");
  constexpr int DZS = wg_stride(COUT), AS = wg_stride(CIN);   // bytes per pixel row in LDS
  constexpr int AW = SEG + 2;
  constexpr int DZ_BYTES = 2 * SEG * DZS, A_BYTES = 4 * AW * AS, BUF_BYTES = DZ_BYTES + A_BYTES;
  constexpr int NDZ = (2 * SEG * (COUT / 8) + 255) / 256, NA = (4 * AW * (CIN / 8) + 255) / 256;
  constexpr int NKS = 2 * (SEG / 16);                        // k-steps of an item
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int is = KSPLIT ? 0 : (wave & 1);
  const int cs0 = KSPLIT ? 0 : 2 * (wave >> 1);
  const int i16 = lane & 15, qrow = i16 >> 2, pq = i16 & 3, chalf = (lane >> 4) & 1;
  const int dz_lane = (8 * h + qrow) * DZS + (cs0 * 32 + 16 * chalf + 4 * pq) * 2;
  const int a_lane = (8 * h + qrow) * AS + (is * 32 + 16 * chalf + 4 * pq) * 2;

  f32x16_t acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][c][i] = 0.f;
  float dbsum[2] = {0.f, 0.f};

  const int nseg = (W + SEG - 1) / SEG, nrp = (H + 1) / 2;
  const long nitems = (long)B * nrp * nseg;
  uint4 sdz[NDZ], sa[NA];
  auto load_item = [&](long item) {       // global -> registers (zeros outside the image)
    const int seg = (int)(item % nseg);
    const long bt = item / nseg;
    const int t0 = 2 * (int)(bt % nrp), b = (int)(bt / nrp);
    const int f0 = seg * SEG;
#pragma unroll
    for (int k = 0; k < NDZ; ++k) {
      const int e = k * 256 + tid;
      const int cg = e % (COUT / 8), p = (e / (COUT / 8)) % SEG, rr = e / ((COUT / 8) * SEG);
      const bool ok = e < 2 * SEG * (COUT / 8) && t0 + rr < H && f0 + p < W;
      const uint4 v = *(const uint4*)(dz + (ok ? (((size_t)b * H + t0 + rr) * W + f0 + p) * dzs_c + cg * 8 : 0));
      sdz[k] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const int e = k * 256 + tid;
      const int cg = e % (CIN / 8), sl = (e / (CIN / 8)) % AW, row = e / ((CIN / 8) * AW);
      const int tt = t0 + row - 1, ff = f0 - 1 + sl;
      const bool ok = e < 4 * AW * (CIN / 8) && tt >= 0 && tt < H && ff >= 0 && ff < W;
      const uint4 v = *(const uint4*)(a + (ok ? (((size_t)b * H + tt) * W + ff) * as_c + cg * 8 : 0));
      sa[k] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto store_item = [&](int buf) {        // registers -> LDS tiles of buffer `buf`
    char* dzb = smem + buf * BUF_BYTES;
    char* ab = dzb + DZ_BYTES;
#pragma unroll
    for (int k = 0; k < NDZ; ++k) {
      const int e = k * 256 + tid;
      const int cg = e % (COUT / 8), p = (e / (COUT / 8)) % SEG, rr = e / ((COUT / 8) * SEG);
      if (e < 2 * SEG * (COUT / 8)) *(uint4*)(dzb + (rr * SEG + p) * DZS + cg * 16) = sdz[k];
    }
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const int e = k * 256 + tid;
      const int cg = e % (CIN / 8), sl = (e / (CIN / 8)) % AW, row = e / ((CIN / 8) * AW);
      if (e < 4 * AW * (CIN / 8)) *(uint4*)(ab + (row * AW + sl) * AS + cg * 16) = sa[k];
    }
  };
  auto tr8 = [&](const char* p0, int stride4) {   // 8 consecutive pixels of the lane's channel: two transposed reads
    const s16x4_t x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0));
    const s16x4_t x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p0 + stride4));
    const uint2 u0 = __builtin_bit_cast(uint2, x0), u1 = __builtin_bit_cast(uint2, x1);
    return make_uint4(u0.x, u0.y, u1.x, u1.y);
  };
  auto kstep = [&](const char* dzb, const char* ab, int rr, int ks) {
    uint4 av[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      av[c] = tr8(dzb + (rr * SEG + ks * 16) * DZS + dz_lane + c * 64, 4 * DZS);
      if (KSPLIT || is == 0) {
        const unsigned u[4] = {av[c].x, av[c].y, av[c].z, av[c].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) dbsum[c] += __uint_as_float(u[e] << 16) + __uint_as_float(u[e] & 0xffff0000u);
      }
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap - dy * 3;
      const uint4 bv = tr8(ab + ((rr + dy) * AW + ks * 16 + dx) * AS + a_lane, 4 * AS);
#pragma unroll
      for (int c = 0; c < 2; ++c)
        acc[tap][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av[c]),
                                                              __builtin_bit_cast(bf16x8_t, bv), acc[tap][c], 0, 0, 0);
    }
  };

  long item = blockIdx.x;
  if (item < nitems) { load_item(item); store_item(0); }
  __syncthreads();
  for (int n = 0; item < nitems; item += gridDim.x, ++n) {
    const long next = item + gridDim.x;
    if (next < nitems) load_item(next);
    const char* dzb = smem + (n & 1) * BUF_BYTES;
    const char* ab = dzb + DZ_BYTES;
    if (KSPLIT) {
#pragma unroll
      for (int j = 0; j < NKS / 4; ++j) {
        const int kidx = wave + 4 * j;
        kstep(dzb, ab, kidx / (SEG / 16), kidx % (SEG / 16));
      }
    } else {
#pragma unroll
      for (int kidx = 0; kidx < NKS; ++kidx) kstep(dzb, ab, kidx / (SEG / 16), kidx % (SEG / 16));
    }
    if (next < nitems) store_item((n + 1) & 1);
    __syncthreads();
  }

  constexpr size_t REC = (size_t)COUT * CIN * 9 + COUT;
  float* out = partial + (KSPLIT ? (size_t)blockIdx.x * 4 + wave : (size_t)blockIdx.x) * REC;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int co = (cs0 + c) * 32 + (i & 3) + 8 * (i >> 2) + 4 * h, ci = is * 32 + r;
        out[((size_t)co * CIN + ci) * 9 + tap] = acc[tap][c][i];
      }
  if (KSPLIT || is == 0) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const float v = dbsum[c] + __shfl_xor(dbsum[c], 32, 64);
      if (h == 0) out[(size_t)COUT * CIN * 9 + (cs0 + c) * 32 + r] = v;
    }
  }
}

// ---- bf16 kernel, third version: two waves per SIMD and asm-pipelined transposed reads.  v2 at one wave per SIMD left
// the compiler's "two reads -> s_waitcnt lgkmcnt(0) -> two MFMAs" schedule fully exposed (and it shuttled accumulators
// between AGPRs and VGPRs): 0.56 PFLOP/s.  Here the workgroup has 8 waves, each owning 9 accumulator tiles (the nine taps
// of one (ci slice, dz slice) pair, 144 VGPRs, no AGPRs), so a k-step is 1 dz + 9 activation fragments for 9 MFMAs; the
// fragment reads (2 x ds_read_b64_tr_b16 each) run PF fragments ahead through inline asm with counted lgkmcnt waits,
// exactly as in conv3x3_mfma.h, in the register regime where that scheme is verified (<= 256 VGPRs, no scratch;
// tools/check_lds_pipeline.py + the bit-identity test against the PIPE = false twin).
//   KS = 1 (64 -> 128): waves = 2 ci slices x 4 dz slices on the same pixels.
//   KS = 4 (32 -> 64):  waves = 2 dz slices x 4 k-step groups; each k-step group writes its own partial record.
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <int OFF, bool PIPE>
__device__ __forceinline__ u32x2_t lds_tr(unsigned addr) {
  u32x2_t v;
  if constexpr (PIPE) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  } else {
    v = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                        (__attribute__((address_space(3))) s16x4_t*)(size_t)(addr + OFF)));
  }
  return v;
}

// The next item's tiles are fetched with buffer loads whose out-of-image / out-of-tile lanes get an out-of-range offset: they
// return zeros without memory traffic and without a branch, so the loads of an item issue back to back with counted waits
// (under exec branches every wait was vmcnt(0)).  Two items in flight (a second register set) were measured with this form:
// 0.684 vs 0.681 ms for the 64 -> 128 layer -- the kernel is not waiting for its prefetch -- and dropped.
#ifdef DFA_STAMPS   // diagnostic build (make stamps): per-wave cycle split of an item, printed by the launcher
static __device__ long long g_diag_wgrad[256 * 8 * 8];
#endif
template <int CIN, int COUT, int SEG, int ROWS, int KS, bool PIPE>
__global__ __launch_bounds__(512, 2) void wgrad3x3_bf16_v3_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ a,
                                                                  float* __restrict__ partial, int B, int H, int W,
                                                                  int dzs_c, int as_c) {
  constexpr int CS = COUT / 32, IS = CIN / 32;
  static_assert(IS * CS * KS == 8, "8 waves = ci slices x dz slices x k-step groups");
  constexpr int DZS = wg_stride(COUT), AS = wg_stride(CIN);   // bytes per pixel row in LDS
  constexpr int AW = SEG + 2;
  constexpr int DZ_BYTES = ROWS * SEG * DZS, A_BYTES = (ROWS + 2) * AW * AS, BUF_BYTES = DZ_BYTES + A_BYTES;
  constexpr int NDZ = (ROWS * SEG * (COUT / 8) + 511) / 512, NA = ((ROWS + 2) * AW * (CIN / 8) + 511) / 512;
  constexpr int NKS = ROWS * (SEG / 16) / KS;                // k-steps per wave per item
  constexpr int PF = 3;                                      // fragments in flight
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int is = wave % IS, cs = (wave / IS) % CS, kg = wave / (IS * CS);
  const int i16 = lane & 15, qrow = i16 >> 2, pq = i16 & 3, chalf = (lane >> 4) & 1;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned dz_lane = lds0 + (8 * h + qrow) * DZS + (cs * 32 + 16 * chalf + 4 * pq) * 2;
  const unsigned a_lane = lds0 + DZ_BYTES + (8 * h + qrow) * AS + (is * 32 + 16 * chalf + 4 * pq) * 2;

  f32x16_t acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  float dbsum = 0.f;

  const int nseg = (W + SEG - 1) / SEG, nrp = (H + ROWS - 1) / ROWS;
  const long nitems = (long)B * nrp * nseg;
  // The next item's tiles come through bounds-checked buffer loads (out-of-image / out-of-tile lanes get an out-of-range
  // offset: zeros without traffic or branches).  PIPE: the NP = NDZ + NA loads of a thread are asm statements spread evenly
  // through the MFMA stream of the current item -- issued together in front of it they sat in the CU's memory pipeline's
  // queue and BLOCKED their waves for 1556 of an item's 4861 cycles (stamps build), with the memory system then idle
  // through compute, LDS store and barrier.  The compiler sees the asm outputs as ready at once; the s_waitcnt vmcnt(0)
  // in front of store_item carries them as "+v" operands (same contract as the LDS fragment reads).
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  constexpr int NP = NDZ + NA;
  u32x4_t sdz[NDZ], sa[NA];
  constexpr unsigned OOR = 0xfffffff0u;              // beyond every buffer: the load returns zeros
  const unsigned dz_bytes = (unsigned)((size_t)H * W * dzs_c * 2), a_bytes = (unsigned)((size_t)H * W * as_c * 2);
  // buffer descriptors of the item being fetched (wave-uniform): {base lo, base hi, bytes, raw-buffer flags}; the asm form
  // takes them as SGPR quads, the compiler-scheduled twin as the builtin's resource type
  i32x4_t rdz = {0, 0, 0, 0}, ra = {0, 0, 0, 0};
  const bf16_t *pdz = dz, *pa = a;
  unsigned live_dz = 0, live_a = 0;
  int ld_t0 = 0, ld_f0 = 0;
  // (item -> (image, row group, segment) by carried counters: the four 64-bit divisions this replaces cost 700 of an item's
  //  8000 cycles)
  const int gstep = (int)gridDim.x;
  const int d_seg = gstep % nseg, d_rp = (gstep / nseg) % nrp, d_b = gstep / (nseg * nrp);
  int c_seg = (int)(blockIdx.x % nseg), c_rp = (int)((blockIdx.x / nseg) % nrp), c_b = (int)(blockIdx.x / (nseg * nrp));
  auto setup_item = [&](long item_) {                // called for items blockIdx.x, + gridDim.x, ... in order
    const bool live = item_ < nitems;                 // past the end: every lane out of range, nothing is fetched
    ld_t0 = ROWS * c_rp;
    ld_f0 = c_seg * SEG;
    const int b = live ? c_b : 0;
    c_seg += d_seg;
    if (c_seg >= nseg) { c_seg -= nseg; ++c_rp; }
    c_rp += d_rp;
    if (c_rp >= nrp) { c_rp -= nrp; ++c_b; }
    c_b += d_b;
    pdz = dz + (size_t)b * H * W * dzs_c;
    pa = a + (size_t)b * H * W * as_c;
    live_dz = live ? dz_bytes : 0;
    live_a = live ? a_bytes : 0;
    if constexpr (PIPE) {
      const unsigned long long udz = (unsigned long long)pdz, ua = (unsigned long long)pa;
      rdz = i32x4_t{__builtin_amdgcn_readfirstlane((int)(unsigned)udz), __builtin_amdgcn_readfirstlane((int)(unsigned)(udz >> 32) & 0xffff),
                    __builtin_amdgcn_readfirstlane((int)live_dz), 0x00020000};
      ra = i32x4_t{__builtin_amdgcn_readfirstlane((int)(unsigned)ua), __builtin_amdgcn_readfirstlane((int)(unsigned)(ua >> 32) & 0xffff),
                   __builtin_amdgcn_readfirstlane((int)live_a), 0x00020000};
    }
  };
  auto buf_load = [&](u32x4_t& dst, bool is_dz, unsigned off) {
    if constexpr (PIPE) {
      if (is_dz) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(off), "s"(rdz));
      else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(off), "s"(ra));
    } else {
      const __amdgpu_buffer_rsrc_t r = is_dz ? __builtin_amdgcn_make_buffer_rsrc((void*)pdz, 0, live_dz, 0x00020000)
                                             : __builtin_amdgcn_make_buffer_rsrc((void*)pa, 0, live_a, 0x00020000);
      dst = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
    }
  };
  auto issue_piece = [&](auto q_c) {                 // piece q of the item set up last: global -> registers
    constexpr int q = decltype(q_c)::value;
    if constexpr (q < NDZ) {
      const int e = q * 512 + tid;
      const int cg = e % (COUT / 8), p = (e / (COUT / 8)) % SEG, rr = e / ((COUT / 8) * SEG);
      // (bitwise &, offset computed on every lane: `&&` became EXEC-masked branches inside the asm-read window)
      const bool ok = (e < ROWS * SEG * (COUT / 8)) & (ld_t0 + rr < H) & (ld_f0 + p < W);
      const unsigned off = (unsigned)((((ld_t0 + rr) * W + ld_f0 + p) * dzs_c + cg * 8) * 2);
      buf_load(sdz[q], true, ok ? off : OOR);
    } else if constexpr (q < NP) {
      constexpr int k = q - NDZ;
      const int e = k * 512 + tid;
      const int cg = e % (CIN / 8), sl = (e / (CIN / 8)) % AW, row = e / ((CIN / 8) * AW);
      const int tt = ld_t0 + row - 1, ff = ld_f0 - 1 + sl;
      const bool ok = (e < (ROWS + 2) * AW * (CIN / 8)) & ((unsigned)tt < (unsigned)H) & ((unsigned)ff < (unsigned)W);
      const unsigned off = (unsigned)(((tt * W + ff) * as_c + cg * 8) * 2);
      buf_load(sa[k], false, ok ? off : OOR);
    }
  };
  auto wait_loads = [&]() {                          // every piece has landed (and the compiler knows the registers changed)
    if constexpr (PIPE) {
      static_assert(NDZ <= 4 && NA <= 4, "operand list of the wait below");
      if constexpr (NDZ == 2 && NA == 3)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(sdz[0]), "+v"(sdz[1]), "+v"(sa[0]), "+v"(sa[1]), "+v"(sa[2]));
      else if constexpr (NDZ == 4 && NA == 4)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(sdz[0]), "+v"(sdz[1]), "+v"(sdz[2]), "+v"(sdz[3]), "+v"(sa[0]), "+v"(sa[1]), "+v"(sa[2]), "+v"(sa[3]));
      else if constexpr (NDZ == 4 && NA == 3)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(sdz[0]), "+v"(sdz[1]), "+v"(sdz[2]), "+v"(sdz[3]), "+v"(sa[0]), "+v"(sa[1]), "+v"(sa[2]));
      else
        static_assert(NDZ == 0, "add the operand list for this tile shape");
    }
  };
  auto store_item = [&](int buf) {                   // registers -> LDS tiles of buffer `buf`
    char* dzb = smem + buf * BUF_BYTES;
    char* ab = dzb + DZ_BYTES;
#pragma unroll
    for (int k = 0; k < NDZ; ++k) {
      const int e = k * 512 + tid;
      const int cg = e % (COUT / 8), p = (e / (COUT / 8)) % SEG, rr = e / ((COUT / 8) * SEG);
      if (e < ROWS * SEG * (COUT / 8)) *(u32x4_t*)(dzb + (rr * SEG + p) * DZS + cg * 16) = sdz[k];
    }
#pragma unroll
    for (int k = 0; k < NA; ++k) {
      const int e = k * 512 + tid;
      const int cg = e % (CIN / 8), sl = (e / (CIN / 8)) % AW, row = e / ((CIN / 8) * AW);
      if (e < (ROWS + 2) * AW * (CIN / 8)) *(u32x4_t*)(ab + (row * AW + sl) * AS + cg * 16) = sa[k];
    }
  };

  // PIPE: the pieces are also WRITTEN to the other LDS buffer inside the MFMA stream (last quarter of the item), so the only
  // serial part of an item is its barrier.  The other buffer was last read in the previous item, behind that item's barrier;
  // piece q has landed once vmcnt <= NP-1-q (loads return in order).  Lanes past the end of a tile write to a spare 1 KiB
  // behind the two buffers instead of branching (the transposed reads need EXEC all ones and no branch in their window).
  const unsigned lds_dummy = lds0 + 2 * BUF_BYTES + lane * 16;
  auto write_piece = [&sdz, &sa, lds0, lds_dummy, tid](auto q_c, int buf) {
    constexpr int q = decltype(q_c)::value;
    const unsigned base = lds0 + buf * BUF_BYTES;
    if constexpr (q < NDZ) {
      const int e = q * 512 + tid;
      const int cg = e % (COUT / 8), p = (e / (COUT / 8)) % SEG, rr = e / ((COUT / 8) * SEG);
      const unsigned addr = e < ROWS * SEG * (COUT / 8) ? base + (rr * SEG + p) * DZS + cg * 16 : lds_dummy;
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(sdz[q]) : "n"(NP - 1 - q));
      asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(sdz[q]) : "memory");
    } else {
      constexpr int k = q - NDZ;
      const int e = k * 512 + tid;
      const int cg = e % (CIN / 8), sl = (e / (CIN / 8)) % AW, row = e / ((CIN / 8) * AW);
      const unsigned addr = e < (ROWS + 2) * AW * (CIN / 8) ? base + DZ_BYTES + (row * AW + sl) * AS + cg * 16 : lds_dummy;
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(sa[k]) : "n"(NP - 1 - q));
      asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(sa[k]) : "memory");
    }
  };

  // one item: NKS k-steps x (1 dz + 9 activation fragments), pipelined PF fragments deep.  Fragment n of k-step j:
  // n = 0 -> dz, n = 1..9 -> tap n-1.  The wave's k-steps are kidx = kg*NKS + j (row kidx / (SEG/16), 16-pixel group).
  auto compute = [&](int buf) {
    constexpr int NF = 10 * NKS;
    const unsigned dzb = dz_lane + buf * BUF_BYTES, ab = a_lane + buf * BUF_BYTES;
    const unsigned kg_dz = (unsigned)(kg * NKS / (SEG / 16) * SEG + (kg * NKS % (SEG / 16)) * 16) * DZS;   // KS > 1: the group's first k-step
    const unsigned kg_a = (unsigned)(kg * NKS / (SEG / 16) * AW + (kg * NKS % (SEG / 16)) * 16) * AS;
    u32x2_t f0[PF], f1[PF];
    uint4 av = make_uint4(0u, 0u, 0u, 0u);
    // PIPE schedule of the next item's NP pieces over the NF fragment steps: global loads at steps 0, G, 2G, .. (first quarter),
    // LDS writes at steps W0, W0+G, .. (last quarter); a write at step t goes out in front of step t's fragment reads, so the
    // counted lgkmcnt wait of fragment c at step s also counts the writes of steps c+1 .. s
    constexpr int G = NF / (4 * NP) < 1 ? 1 : NF / (4 * NP);
    constexpr int W0 = NF - NP * G;
    static_assert(2 * NP * G <= NF, "loads and writes of the pieces overlap");
    auto is_wstep = [](int t) constexpr { return t >= W0 && t < NF && (t - W0) % G == 0; };
    const int nbuf = buf ^ 1;
    auto step = [&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      if constexpr (PIPE && s % G == 0 && s / G < NP) issue_piece(std::integral_constant<int, s / G>{});
      if constexpr (PIPE && is_wstep(s)) write_piece(std::integral_constant<int, (s - W0) / G>{}, nbuf);
      if constexpr (s < NF) {
        constexpr int j = s / 10, n = s % 10;
        constexpr int rr = j / (SEG / 16), ks = j % (SEG / 16);     // relative to the group's first k-step (NKS <= SEG/16 or KS == 1)
        if constexpr (n == 0) {
          constexpr int off = (rr * SEG + ks * 16) * DZS;
          f0[s % PF] = lds_tr<off, PIPE>(dzb + kg_dz);
          f1[s % PF] = lds_tr<off + 4 * DZS, PIPE>(dzb + kg_dz);
        } else {
          constexpr int tap = n - 1, dy = tap / 3, dx = tap % 3;
          constexpr int off = ((rr + dy) * AW + ks * 16 + dx) * AS;
          f0[s % PF] = lds_tr<off, PIPE>(ab + kg_a);
          f1[s % PF] = lds_tr<off + 4 * AS, PIPE>(ab + kg_a);
        }
      }
      if constexpr (s >= PF - 1) {
        constexpr int c = s - (PF - 1);
        constexpr int n = c % 10;
        constexpr int nw = []() constexpr { int k = 0; for (int t = c + 1; t <= s; ++t) k += (t >= W0 && t < NF && (t - W0) % G == 0) ? 1 : 0; return k; }();
        constexpr int young = 2 * ((NF - 1 - c) < (PF - 1) ? (NF - 1 - c) : (PF - 1)) + nw;
        static_assert(young <= 15, "lgkmcnt is a 4-bit counter");
        if constexpr (PIPE) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f0[c % PF]), "+v"(f1[c % PF]) : "n"(young));
        const uint4 fv = make_uint4(f0[c % PF][0], f0[c % PF][1], f1[c % PF][0], f1[c % PF][1]);
        if constexpr (n == 0) {
          av = fv;
          {   // every wave sums its dz fragment (only the ci-slice-0 waves write the result): a wave-uniform `if` here is a
              // branch inside the asm-read window, and the compiler is free to lay its block out of line
            const unsigned u[4] = {fv.x, fv.y, fv.z, fv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) dbsum += __uint_as_float(u[e] << 16) + __uint_as_float(u[e] & 0xffff0000u);
          }
        } else {
          acc[n - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, fv),
                                                               acc[n - 1], 0, 0, 0);
        }
      }
    };
    static_for(std::make_integer_sequence<int, NF + PF - 1>{}, step);
  };
  static_assert(KS == 1 || NKS <= SEG / 16, "a k-step group stays inside one row");

#ifdef DFA_STAMPS
  long long seg_[5] = {0, 0, 0, 0, 0};
  long long t_prev = __builtin_amdgcn_s_memtime();
  const long long t_begin = t_prev, r_begin = __builtin_amdgcn_s_memrealtime();
  auto stamp = [&](int k) { const long long t = __builtin_amdgcn_s_memtime(); seg_[k] += t - t_prev; t_prev = t; };
  int n_items = 0;
#else
  auto stamp = [&](int) {};
#endif
  auto issue_all = [&]() { static_for(std::make_integer_sequence<int, NP>{}, issue_piece); };
  long item = blockIdx.x;
  if (item < nitems) { setup_item(item); issue_all(); wait_loads(); store_item(0); }
  __syncthreads();
  stamp(4);
  for (int n = 0; item < nitems; item += gridDim.x, ++n) {
    const long next = item + gridDim.x;
    setup_item(next);                          // past the end: every lane out of range, nothing is fetched
    if constexpr (!PIPE) issue_all();
    stamp(0);
    compute(n & 1);                            // (PIPE: loads the next item's pieces and writes them to the other buffer on the way)
    stamp(1);
    if constexpr (PIPE) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the asm LDS writes, which the compiler does not count
    } else {
      wait_loads();
      if (next < nitems) store_item((n + 1) & 1);
    }
    stamp(2);
    __syncthreads();
    stamp(3);
#ifdef DFA_STAMPS
    ++n_items;
#endif
  }
#ifdef DFA_STAMPS
  if (lane == 0 && blockIdx.x < 256) {
    long long* dd = g_diag_wgrad + ((size_t)blockIdx.x * 8 + wave) * 8;
    for (int k = 0; k < 5; ++k) dd[k] = seg_[k];
    dd[5] = __builtin_amdgcn_s_memtime() - t_begin;
    dd[6] = __builtin_amdgcn_s_memrealtime() - r_begin;
    dd[7] = n_items;
  }
#endif

  // The record is written in ACCUMULATOR order, [tap][g][wave of the record][lane][4]: one 16-byte store per lane and register quad,
  // 1 KiB contiguous per wave instruction (dW order [co][ci][tap] meant 4-byte stores 36 bytes apart: 144 scattered store
  // instructions per wave, ~30 us of every launch); reduce_wgrad_record_kernel (perm = 1) undoes the permutation while it sums.
  constexpr size_t REC = (size_t)COUT * CIN * 9 + COUT;
  constexpr int NWR = IS * CS;
  float* out = partial + ((size_t)blockIdx.x * KS + kg) * REC;
  const int wrec = is + IS * cs;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *(float4*)(out + ((size_t)((tap * 4 + g) * NWR + wrec) * 64 + lane) * 4) =
          make_float4(acc[tap][4 * g], acc[tap][4 * g + 1], acc[tap][4 * g + 2], acc[tap][4 * g + 3]);
  if (is == 0) {
    const float v = dbsum + __shfl_xor(dbsum, 32, 64);
    if (h == 0) out[(size_t)COUT * CIN * 9 + cs * 32 + r] = v;
  }
}

template <int CIN, int COUT, int SEG, int ROWS, int KS, bool PIPE>
static hipError_t launch_wgrad_bf16_v3(const void* dz, const void* a, float* partial, int B, int H, int W, int nwg,
                                       hipStream_t s, int dzs_c, int as_c) {
  constexpr int LDS = 2 * (ROWS * SEG * wg_stride(COUT) + (ROWS + 2) * (SEG + 2) * wg_stride(CIN)) + 1024;   // + spare write slots
  static_assert(LDS <= 160 * 1024, "LDS of a CU");
  auto kern = wgrad3x3_bf16_v3_kernel<CIN, COUT, SEG, ROWS, KS, PIPE>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(512), LDS, s, (const bf16_t*)dz, (const bf16_t*)a, partial, B, H, W, dzs_c, as_c);
#ifdef DFA_STAMPS
  {
    static int calls = 0;
    if (++calls == 30 && PIPE) {
      static long long hbuf[256 * 8 * 8];
      (void)hipDeviceSynchronize();
      (void)hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(g_diag_wgrad), sizeof(hbuf));
      const int nw = (nwg < 256 ? nwg : 256) * 8;
      double m[7] = {0, 0, 0, 0, 0, 0, 0}, items = 0;
      for (int i = 0; i < nw; ++i) { for (int k = 0; k < 7; ++k) m[k] += hbuf[i * 8 + k]; items += hbuf[i * 8 + 7]; }
      fprintf(stderr, "[stamps wgrad v3<%d,%d,seg %d,rows %d,ks %d>] cycles per wave-item: load issue %.0f  compute %.0f  store to LDS %.0f  "
                      "barrier %.0f  | prologue/item %.0f  lifetime/item %.0f  clock %.3f GHz  items/wave %.1f\n", CIN, COUT, SEG, ROWS, KS,
              m[0] / items, m[1] / items, m[2] / items, m[3] / items, m[4] / items, m[5] / items, m[5] / (m[6] * 10.0), items / nw);
    }
  }
#endif
  return hipGetLastError();
}

template <int CIN, int COUT, int SEG, bool KSPLIT>
static hipError_t launch_wgrad_bf16_v2(const void* dz, const void* a, float* partial, int B, int H, int W, int nwg,
                                       hipStream_t s, int dzs_c, int as_c) {
  constexpr int LDS = 2 * (2 * SEG * wg_stride(COUT) + 4 * (SEG + 2) * wg_stride(CIN));
  auto kern = wgrad3x3_bf16_v2_kernel<CIN, COUT, SEG, KSPLIT>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), LDS, s, (const bf16_t*)dz, (const bf16_t*)a, partial, B, H, W, dzs_c, as_c);
  return hipGetLastError();
}

template <typename T, int CIN, int COUT>
static hipError_t launch_wgrad_t(const void* dz, const void* a, float* partial, int B, int H, int W, int nwg,
                                 hipStream_t s, int dzs_c, int as_c) {
  constexpr int LDS = (WG_SEG * COUT + 3 * (WG_SEG + 2) * CIN) * 4;
  auto kern = wgrad3x3_mfma_kernel<T, CIN, COUT>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), LDS, s, (const T*)dz, (const T*)a, partial, B, H, W, dzs_c, as_c);
  return hipGetLastError();
}

static int g_wgrad_variant = 3;   // 3 = asm-pipelined v3 (product), 30 = its compiler-scheduled twin, 2 = v2 (test hooks)
void set_wgrad_variant(int v) { g_wgrad_variant = v; }
static int wgrad_variant() { return g_wgrad_variant; }

// dW [COUT][CIN][3][3] and db [COUT] <- dz [B][H][W][COUT], a [B][H][W][CIN]; partial: nwg * (128*64*9 + 256) floats
// (the bf16 32 -> 64 kernel writes 4 records of 64*32*9 + 64 floats per workgroup)
hipError_t launch_wgrad3x3(int prec, int cin, int cout, const void* dz, const void* a, float* partial, float* dw,
                           float* db, int B, int H, int W, int nwg, hipStream_t s) {
  return launch_wgrad3x3_window(prec, cin, cout, cin, cout, 0, 0, dz, a, partial, dw, db, B, H, W, nwg, s);
}

// channel-window form: the kernel shapes are (cin, cout) in {(32,64), (64,128)}; a layer with more channels is covered by
// several launches, each on the window [ci_off, ci_off+cin) x [co_off, co_off+cout) of tensors with cin_total / cout_total
// channels per pixel.  dw is the FULL [cout_total][cin_total][9] gradient; db (may be null) the full [cout_total] one.
hipError_t launch_wgrad3x3_window(int prec, int cin, int cout, int cin_total, int cout_total, int ci_off, int co_off,
                                  const void* dz, const void* a, float* partial, float* dw, float* db, int B, int H,
                                  int W, int nwg, hipStream_t s) {
  const size_t es = (prec == DFA_PREC_BF16) ? 2 : 4;
  const void* dzw = (const char*)dz + (size_t)co_off * es;
  const void* aw = (const char*)a + (size_t)ci_off * es;
  hipError_t e;
  int nparts = nwg;     // partial records written
  if (prec == DFA_PREC_BF16) {
    const int variant = wgrad_variant();
    if (cin == 64 && cout == 128) {
      if (variant == 2) e = launch_wgrad_bf16_v2<64, 128, 32, false>(dzw, aw, partial, B, H, W, nwg, s, cout_total, cin_total);
      else if (variant == 30) e = launch_wgrad_bf16_v3<64, 128, 32, 4, 1, false>(dzw, aw, partial, B, H, W, nwg, s, cout_total, cin_total);
      else e = launch_wgrad_bf16_v3<64, 128, 32, 4, 1, true>(dzw, aw, partial, B, H, W, nwg, s, cout_total, cin_total);
    } else if (cin == 32 && cout == 64) {
      if (variant == 2) e = launch_wgrad_bf16_v2<32, 64, 64, true>(dzw, aw, partial, B, H, W, nwg, s, cout_total, cin_total);
      else if (variant == 30) e = launch_wgrad_bf16_v3<32, 64, 64, 4, 4, false>(dzw, aw, partial, B, H, W, nwg, s, cout_total, cin_total);
      else e = launch_wgrad_bf16_v3<32, 64, 64, 4, 4, true>(dzw, aw, partial, B, H, W, nwg, s, cout_total, cin_total);
      nparts = 4 * nwg;
    }
    else return hipErrorInvalidValue;
  } else {
    if (cin == 64 && cout == 128) e = launch_wgrad_t<float, 64, 128>(dzw, aw, partial, B, H, W, nwg, s, cout_total, cin_total);
    else if (cin == 32 && cout == 64) e = launch_wgrad_t<float, 32, 64>(dzw, aw, partial, B, H, W, nwg, s, cout_total, cin_total);
    else return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  const int n = cout * cin * 9;
  // weight block: partial record [cout][cin][9] -> dw rows co_off.., columns ci_off.. of [cout_total][cin_total][9]
  const int perm = (prec == DFA_PREC_BF16 && wgrad_variant() != 2) ? 1 : 0;   // the v3 kernels write accumulator-order records
  return launch_reduce_wgrad_record(partial, nparts, n + cout, cin, cout, cin_total, ci_off, co_off, dw, db, s, perm);
}

}  // namespace dfa
