// conv3x3_mfma.h -- 3x3 / pad 1 convolution as an implicit GEMM on the gfx950 matrix cores, with the
// BatchNorm(eval, folded) + ReLU + pooling / time-mean epilogue fused in.
//
// Replaces, per launch, one "Conv2d -> BatchNorm2d -> ReLU [-> AvgPool2d]" group of the reference:
//   src/model.py:21-25 (block 2), :27-29 + :37 (block 3 + mean over T),
//   src/model_cae.py:40-55 (encoder blocks 2-4); in train mode (EPI_PLAIN) it stores the pre-BatchNorm output
//   and serves as the data-gradient convolution of loss.backward() (src/train.py:75).
//
// Data layout (HBM): activations are channels-last  act[b][t][f][c]  so that one tap of one pixel is
// CIN contiguous elements; weights are pre-packed by pack.hip in the exact register order the MFMA
// operand wants:  wpack[cout/32][tap][kgroup][lane] (16 bytes each).
//
// Work decomposition
//   workgroup  = (utterance b, strip of 32 feature columns, chunk of 32*NSL output channels); it walks
//                DOWN the time axis keeping a ring of input rows in LDS, so every input element is read from
//                HBM once per strip (+2 halo columns) and there is no vertical halo re-read.
//   wave       = (N-slice nsl of 32 output channels, M-group mg).  The wave keeps its full 9 x CIN x 32 weight
//                slice in VGPRs for the whole kernel, and streams activation fragments from the LDS ring.
//   iteration  = 2*RP output rows (RP row pairs).  A "unit" is one pair of 32-pixel tiles (rows t, t+1; same
//                32 columns): both accumulators share the fragments of the two input rows they have in
//                common, so a unit issues 12*NKG ds_read_b128 for 18*NKG k-groups of MFMA work.
//   staging    = ring block j holds input rows [2RP*j-1, 2RP*(j+1)-1).  Iteration `it` computes from
//                blocks it, it+1 while block it+2 is fetched (through registers, or by LDS-DMA), one
//                __syncthreads() per iteration.
//
// Instruction economy (rocprofv3 showed ~5 VALU per MFMA in the first version; the matrix pipe shares issue slots
// with the VALU):
//   * LDS image: pixel slot s of ring row q owns PB = CIN*sizeof(T) bytes at (q*SP + s)*PB; 16-byte chunk c sits at
//     physical chunk c ^ swz(s).  The swizzle depends on the COLUMN only and SP*PB is a multiple of 256 bytes, so
//     (a) the 16 lanes of every ds_read_b128 lane group (consecutive columns, same logical chunk) hit 16 distinct
//     bank slots, and (b) a lane's byte offset inside a row is loop-invariant: it is computed once, the k-group is
//     one XOR with an immediate, and -- the iteration loop being unrolled over the ring period 3 -- the row offset
//     is an instruction immediate.
//   * staging addresses are per-thread constants plus a per-iteration scalar.
//   * MFMA operands are swapped: weights are the A operand, activations the B operand, so the accumulator holds
//     pixel = lane&31 and channels (i&3)+8*(i>>2)+4*(lane>>5) in its 16 registers.  Bias is the accumulator's initial
//     value; every lane owns 4 consecutive channels x 4 groups of its pixel, which become 16-byte stores (bf16: after
//     one v_permlane32_swap per dword, the T21 idiom of the CDNA guide); the time-mean epilogue writes whole rows of
//     the embedding with no transpose.
//   * average pooling's 1/2 (1/4) is folded into the packed weights and bias (relu(s*x) = s*relu(x), s > 0, exact).
//
// MFMA: bf16 -> v_mfma_f32_32x32x16_bf16 (8 bf16 per lane = one 16-byte chunk);
//       f32  -> v_mfma_f32_32x32x2_f32  (exact fp32 fma chain; one 16-byte chunk feeds 4 MFMAs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <type_traits>
#include <utility>

#include "rng.h"

namespace dfa {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct bf16_t {
  unsigned short v;
};

__device__ __forceinline__ float bf16_to_float(bf16_t x) { return __uint_as_float(((unsigned)x.v) << 16); }
__device__ __forceinline__ bf16_t float_to_bf16(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
  bf16_t r;
  r.v = __builtin_bit_cast(unsigned short, b);
  return r;
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  bf16x2_t v = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(unsigned, v);
}
template <typename T>
__device__ __forceinline__ T cvt_out(float f);
template <>
__device__ __forceinline__ float cvt_out<float>(float f) { return f; }
template <>
__device__ __forceinline__ bf16_t cvt_out<bf16_t>(float f) { return float_to_bf16(f); }

// ReLU as ONE instruction: v_med3_f32(x, 0, lim) with lim = +inf held in an SGPR the compiler cannot see through
// (relu_limit()).  fmaxf(x, 0) costs two -- the compiler canonicalises the operand first -- and so does a med3 against a
// literal +inf, which it folds back into that max.  Not inline asm either: the MFMA -> VALU hazard nops do not cover asm.
__device__ __forceinline__ float relu_limit() {
  float lim = __builtin_inff();
  asm volatile("" : "+s"(lim));
  return lim;
}
__device__ __forceinline__ float relu1(float x, float lim) { return __builtin_amdgcn_fmed3f(x, 0.f, lim); }

enum { EPI_POOL_H2 = 0, EPI_POOL_2X2 = 1, EPI_MEAN_T = 2, EPI_PLAIN = 3, EPI_RAW = 4 };

struct ConvArgs {
  const void* in;      // [B][H][W][CIN] T
  const uint4* wpack;  // [COUT/32][9][CIN/KG][64] x 16 bytes
  const float* bias;   // [COUT] folded bias
  void* out;           // POOL_H2: [B][H/2][W][COUT]; POOL_2X2: [B][H/2][W/2][COUT]; PLAIN: [B][H][W][COUT]
  float* emb;          // MEAN_T: [B][COUT][W] fp32 (= mean over H)
  int B, H, W, COUT;
  int nstrips;
  float inv_h;
  int relu;            // PLAIN only: apply ReLU (1) or not (0)
  // K-split support (Cin larger than one launch can keep in registers, e.g. fp32 Cin = 128): a launch may read a
  // CIN-channel window of wider pixels, start from previously stored partial sums and/or store raw partial sums.
  int in_pix_bytes;    // bytes between consecutive input pixels (0 = CIN*sizeof(T), i.e. dense)
  int in_ch_off_bytes; // byte offset of this launch's first input channel inside a pixel
  const float* acc_in; // ACCIN: [B][H][W][COUT] fp32 partial sums to start from (they already contain the bias)
  float* raw_out;      // EPI_RAW: [B][H][W][COUT] fp32 partial sums (bias included, no activation)
  // train-mode BatchNorm statistics (EPI_PLAIN only): per-workgroup partial sums of the stored values v and v*v
  // per output channel, partial[(blockIdx.x * COUT + channel) * 2 + {0,1}]; reduced in a fixed order by bn_finalize.
  float* stats_partial;
  const void* zero_page;  // DMA staging: >= 16 zero bytes in device memory (source of out-of-image chunks)
  // time-axis split for small batches (conv3_m16.hip, conv_split.hip): blockIdx.z = segment, a segment walks seg_iters
  // iterations (0 = the whole axis).  The time mean is ALWAYS summed in canonical chunks of chunk_iters iterations (a
  // multiple of 6 that depends on H only), the chunk sums added in chunk order: an unsplit workgroup keeps the running total
  // in LDS, a split one writes every chunk sum (unscaled) to emb + chunk * emb_seg_stride floats and the classifier kernel
  // adds them in the same order -- so logits do not depend on the batch size or on the split, bit for bit.
  int seg_iters, chunk_iters;
  size_t emb_seg_stride;
  // conv_split.hip PLAIN_BF16 (data gradient of block 2): thresh != 0 zeroes the elements the forward's dropout layer dropped
  // (same Philox draw, element index = output index) -- the keep mask of the pooled a1 applied where da1 is produced
  DropCfg drop;
  // held-clock probe (conv3_m16.hip eval form; context option "clock_probe"): when non-null, lane 0 of the first 1024 workgroups
  // (blockIdx.y = z = 0) writes {delta s_memtime (shader cycles), delta s_memrealtime (100 MHz ticks)} around its main loop to
  // clock_stamps[2 * blockIdx.x ..]; never read by any kernel.  Null (the default): two scalar compares per workgroup.
  long long* clock_stamps;
};

// chunk swizzle as a function of the pixel slot (column) only
template <int PB>
__device__ __forceinline__ int lds_swz(int slot) {
  if (PB == 64) return (slot >> 2) & 3;
  if (PB == 128) return (slot >> 1) & 7;
  return slot & 15;  // 256, 512, 1024
}

// acc += W(A operand: 32 channels x K) . X(B operand: K x 32 pixels)
template <typename T>
struct Mma;
template <>
struct Mma<bf16_t> {
  static __device__ __forceinline__ f32x16_t run(const uint4& w, const uint4& x, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), c,
                                                   0, 0, 0);
  }
};
template <>
struct Mma<float> {
  static __device__ __forceinline__ f32x16_t run(const uint4& w, const uint4& x, f32x16_t c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.x), __uint_as_float(x.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.y), __uint_as_float(x.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.z), __uint_as_float(x.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.w), __uint_as_float(x.w), c, 0, 0, 0);
    return c;
  }
};

// ---- explicit LDS fragment pipeline.  hipcc schedules "ds_read_b128 -> s_waitcnt lgkmcnt(0) -> MFMA" with ONE fragment
// buffer (every MFMA pair eats the full LDS latency), so the fragment reads are issued through inline asm PFD reads
// ahead of their use and retired with counted waits.  LDS operations complete in order, so "lgkmcnt(N)" guarantees
// everything older than the N youngest reads has landed; reads the compiler adds on its own only make a wait stricter.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
template <int OFF, bool PIPE>
__device__ __forceinline__ u32x4_t lds_frag(unsigned addr) {
  u32x4_t v;
  if constexpr (PIPE) {
    constexpr int LO = OFF & 0xFFFF, HI = OFF - LO;   // the DS offset field is 16 bits
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr + HI), "n"(LO));
  } else {   // compiler-scheduled read (it places its own s_waitcnt)
    v = *(const u32x4_t*)((const __attribute__((address_space(3))) char*)(size_t)(addr + OFF));
  }
  return v;
}
template <int N>
__device__ __forceinline__ void lds_wait(u32x4_t& v) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait4(u32x4_t& a, u32x4_t& b, u32x4_t& c, u32x4_t& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
template <int... Is, typename F>
__device__ __forceinline__ void static_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

template <typename T, int CIN, int NSL, int MG, int RP, int EPI>
struct ConvCfg {
  static constexpr int ES = sizeof(T);
  static constexpr int KG = 32 / ES;       // input channels per k-group (two 16-byte chunks)
  static constexpr int NKG = CIN / KG;
  static constexpr int PB = CIN * ES;      // bytes per pixel
  static constexpr int CPP = PB / 16;      // 16-byte chunks per pixel
  static constexpr int SLOTS = 34;         // 32 columns + 2 halo
  static constexpr int SPA = (PB >= 256) ? 1 : 256 / PB;          // row pitch must make SP*PB a multiple of 256 B
  static constexpr int SP = (SLOTS + SPA - 1) / SPA * SPA;
  static constexpr int BR = 2 * RP;        // rows per ring block
  static constexpr int NT = 64 * NSL * MG;
  static constexpr int NCH = BR * SP * CPP;  // physical chunks per ring block
  static constexpr int NLD = (NCH + NT - 1) / NT;
  static constexpr int UPW = RP / MG;      // units per wave per iteration
  static constexpr int ROW_BYTES = SP * PB;
  static constexpr int RING_BYTES = 3 * BR * ROW_BYTES;
  static constexpr int BIAS_BYTES = NSL * 32 * 4;
  static constexpr int STAT_BYTES = (EPI == EPI_PLAIN) ? MG * NSL * 32 * 2 * 4 : 0;
  static constexpr int LDS_BYTES = RING_BYTES + BIAS_BYTES + STAT_BYTES;
  static_assert(RP % MG == 0, "row pairs must split evenly over the M groups");
  static_assert((BR & (BR - 1)) == 0, "rows per block must be a power of two");
  static_assert(EPI != EPI_MEAN_T || MG == 1, "time-mean epilogue keeps column sums per wave: MG must be 1");
  static_assert(CIN % KG == 0, "CIN must be a multiple of the k-group");
};

#ifdef DFA_STAMPS
static __device__ long long g_diag[2048 * 4 * 8];
#endif
template <typename T, int CIN, int NSL, int MG, int RP, int MT, int EPI, int MINW, bool ACCIN = false, bool DMA = false,
          bool STATS = false, int PFD = -1>
__global__ __launch_bounds__(64 * NSL * MG, MINW) void conv3x3_mfma_kernel(ConvArgs a) {
  static_assert(MT == 1, "strips are 32 columns wide");
  using C = ConvCfg<T, CIN, NSL, MG, RP, EPI>;
  constexpr int PB = C::PB, CPP = C::CPP, SP = C::SP, BR = C::BR, NT = C::NT, NKG = C::NKG, ROWB = C::ROW_BYTES;
  static_assert(!STATS || EPI == EPI_PLAIN, "BatchNorm statistics ride on the PLAIN epilogue");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsl = (NSL == 1) ? 0 : wave % NSL;
  const int mg = (MG == 1) ? 0 : wave / NSL;
  const int r = lane & 31, h = lane >> 5;

  // XCD-aware block order: blocks with equal blockIdx.x % 8 share an XCD (and its L2); hand each XCD a
  // contiguous range of (utterance, strip) ids so the strips that share halo columns meet in one L2.
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, xi = bid >> 3;
  const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
  const int b = logical / a.nstrips;
  const int strip = logical - b * a.nstrips;
  const int f0 = strip * 32;
  const int H = a.H, W = a.W, COUT = a.COUT;
  const int cout_base = blockIdx.y * (NSL * 32);
  const int nb = cout_base + nsl * 32;  // first output channel of this wave's slice

  const int ipb = a.in_pix_bytes ? a.in_pix_bytes : PB;
  const char* in_b = (const char*)a.in + (size_t)b * H * W * ipb + a.in_ch_off_bytes;

  // ---- weights: the wave's [9][NKG] 16-byte fragments stay in registers for the whole kernel
  uint4 w[9][NKG];
  {
    const uint4* wp = a.wpack + ((size_t)(blockIdx.y * NSL + nsl) * 9 * NKG) * 64 + lane;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kg = 0; kg < NKG; ++kg) w[tap][kg] = wp[(tap * NKG + kg) * 64];
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of smem
  const float rlim = relu_limit();
  // PFD > 0: asm-pipelined fragment reads, PFD reads in flight; 0: compiler-scheduled reads; -1: pipelined (depth 4) for
  // the two-waves-per-SIMD bf16 kernels.  The one-wave-per-SIMD kernels keep weights in AGPRs and spill; there the
  // compiler-scheduled form is used (an fp32 ACCIN kernel produced wrong sums with pipelined reads under that register
  // pressure) -- every pipelined instantiation is checked bit-for-bit against its PFD = 0 twin on the GPU.
  constexpr bool PIPE = PFD > 0 || (PFD < 0 && MINW >= 2 && sizeof(T) == 2);
  constexpr int PF = PIPE ? (PFD > 0 ? PFD : 4) : 1;
  constexpr bool EARLY_RELU = PIPE && (EPI == EPI_POOL_H2 || EPI == EPI_POOL_2X2 || EPI == EPI_MEAN_T);
  float* bias_lds = (float*)(smem + C::RING_BYTES);
  if (tid < NSL * 32) bias_lds[tid] = a.bias[cout_base + tid];

  // ---- per-lane fragment offsets inside a ring row (loop-invariant): slot = r + dx, logical chunk 2*kg + h.
  // xa[dx] carries the kg = 0 address; k-group kg is xa[dx] ^ (kg << 5) (see header).
  int xa[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int slot = r + dx, s = lds_swz<PB>(slot);
    xa[dx] = slot * PB + (((h ^ (s & 1)) << 4) | ((s >> 1) << 5));
  }

  // ---- staging constants: thread's k-th PHYSICAL chunk of a ring block (pad slots and out-of-image columns -> zeros)
  int s_off[C::NLD];     // source byte offset relative to row (BR*j - 1), -1 when the column is never valid
#pragma unroll
  for (int k = 0; k < C::NLD; ++k) {
    const int g = k * NT + tid;
    const int rowi = g / (SP * CPP), rem = g - rowi * (SP * CPP);
    const int slot = rem / CPP, cph = rem % CPP;
    const int c = cph ^ lds_swz<PB>(slot);
    const int f = f0 - 1 + slot;
    const bool ok = (g < C::NCH) && (slot < C::SLOTS) && (f >= 0) && (f < W);
    s_off[k] = ok ? (rowi * W + f) * ipb + c * 16 : -1;
  }
  uint4 stg[DMA ? 1 : C::NLD];
  auto stage_load = [&](int j) {  // global -> registers
#pragma unroll
    for (int k = 0; k < C::NLD; ++k) {
      const int t = BR * j - 1 + (k * NT + tid) / (SP * CPP);
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (s_off[k] >= 0 && t >= 0 && t < H) v = *(const uint4*)(in_b + (ptrdiff_t)(BR * j - 1) * W * ipb + s_off[k]);
      stg[DMA ? 0 : k] = v;
    }
  };
  auto stage_store = [&](int ringblk) {  // registers -> LDS (physical chunk order)
#pragma unroll
    for (int k = 0; k < C::NLD; ++k) {
      const int g = k * NT + tid;
      if (g < C::NCH) *(uint4*)(smem + ringblk * BR * ROWB + g * 16) = stg[DMA ? 0 : k];
    }
  };
  // LDS-DMA variant (global_load_lds_dwordx4): no staging VGPRs, no ds_write.  One wave instruction fills 64
  // consecutive PHYSICAL 16-byte chunks (wave-uniform LDS base + lane*16); the swizzle lives in the per-lane SOURCE
  // address.  Out-of-image chunks read a 16-byte zero page.  Retired by the vmcnt(0) of the iteration's barrier.
  auto stage_dma = [&](int j, int ringblk) {
#pragma unroll
    for (int k = 0; k < C::NLD; ++k) {
      const int g = k * NT + tid;
      if (g < C::NCH) {
        const int t = BR * j - 1 + g / (SP * CPP);
        const char* src = (s_off[k] >= 0 && t >= 0 && t < H) ? in_b + (ptrdiff_t)(BR * j - 1) * W * ipb + s_off[k]
                                                             : (const char*)a.zero_page;
        char* dst = smem + ringblk * BR * ROWB + (k * NT + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };

  // ---- epilogue state
  float cs[EPI == EPI_MEAN_T ? 16 : 1];              // MEAN_T: running column sums (pixel = lane, channel = register)
  float st1[STATS ? 16 : 1], st2[STATS ? 16 : 1];    // PLAIN: per-channel sum / sum of squares over this lane's pixels
  if (EPI == EPI_MEAN_T) {
#pragma unroll
    for (int i = 0; i < 16; ++i) cs[i] = 0.f;
  }
  if (STATS) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { st1[i] = 0.f; st2[i] = 0.f; }
  }
  const int col = f0 + r;                    // this lane's output pixel column
  const bool col_ok = col < W;

  const int niter = (H + BR - 1) / BR;
#ifdef DFA_STAMPS
  long long seg[6] = {0, 0, 0, 0, 0, 0};
  long long t_prev = __builtin_amdgcn_s_memtime();
  const long long t_begin = t_prev;
  auto stamp = [&](int k) { const long long t = __builtin_amdgcn_s_memtime(); seg[k] += t - t_prev; t_prev = t; };
#else
  auto stamp = [&](int) {};
#endif
  if (DMA) {
    stage_dma(0, 0);
    stage_dma(1, 1);
  } else {
    stage_load(0);
    stage_store(0);
    stage_load(1);
    stage_store(1);
  }
  if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // one unit: output rows t0 = BR*it + 2*RPI, t0+1; PH = it % 3 (ring phase), both compile-time
  auto unit = [&](auto ph_c, auto rp_c, int it) {
    constexpr int PH = decltype(ph_c)::value, RPI = decltype(rp_c)::value;
    f32x16_t acc0, acc1;
    const int t0 = BR * it + 2 * RPI;
    if (ACCIN) {
      const float* i0 = a.acc_in + (((size_t)b * H + t0) * W + col) * COUT + nb + 4 * h;
      const float* i1 = i0 + (size_t)W * COUT;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
        if (col_ok && t0 < H) v0 = *(const float4*)(i0 + 8 * g);
        if (col_ok && t0 + 1 < H) v1 = *(const float4*)(i1 + 8 * g);
        acc0[4 * g] = v0.x; acc0[4 * g + 1] = v0.y; acc0[4 * g + 2] = v0.z; acc0[4 * g + 3] = v0.w;
        acc1[4 * g] = v1.x; acc1[4 * g + 1] = v1.y; acc1[4 * g + 2] = v1.z; acc1[4 * g + 3] = v1.w;
      }
    }
    constexpr int NR = 12 * NKG;   // fragment reads of a unit, in (row i, dx, kg) order
    constexpr int S_RELU0 = 9 * NKG + (NKG >= 2 ? 2 : 1);         // acc0's last MFMA is consume step 9*NKG - 1
    u32x4_t xb[PF];
    auto step = [&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      if constexpr (s < NR) {
        constexpr int i = s / (3 * NKG), dx = (s / NKG) % 3, kg = s % NKG;   // input row key q = BR*it + 2*RPI + i
        constexpr int ringrow = (BR * PH + 2 * RPI + i) % (3 * BR);
        xb[s % PF] = lds_frag<ringrow * ROWB, PIPE>(lds0 + (xa[dx] ^ (kg << 5)));
      }
      if constexpr (s >= PF - 1) {
        constexpr int c = s - (PF - 1);
        constexpr int i = c / (3 * NKG), dx = (c / NKG) % 3, kg = c % NKG;
        constexpr int young = (NR - 1 - c) < (PF - 1) ? (NR - 1 - c) : (PF - 1);
        if constexpr (PIPE) lds_wait<young>(xb[c % PF]);
        const uint4 xv = __builtin_bit_cast(uint4, xb[c % PF]);
        if constexpr (i <= 2) acc0 = Mma<T>::run(w[i * 3 + dx][kg], xv, acc0);
        if constexpr (i >= 1) acc1 = Mma<T>::run(w[(i - 1) * 3 + dx][kg], xv, acc1);
        if constexpr (EARLY_RELU && c == S_RELU0) {   // rows 0..2 of acc0 are complete: its ReLU hides under acc1's last MFMAs
#pragma unroll
          for (int e = 0; e < 16; ++e) acc0[e] = relu1(acc0[e], rlim);
        }
      }
    };
    if (!ACCIN) {  // bias is the accumulator's initial value (EPI_RAW partial sums carry it into the ACCIN launch)
      const unsigned ba = lds0 + C::RING_BYTES + (nsl * 32 + 4 * h) * 4;
      u32x4_t b0 = lds_frag<0, PIPE>(ba), b1 = lds_frag<32, PIPE>(ba), b2 = lds_frag<64, PIPE>(ba), b3 = lds_frag<96, PIPE>(ba);
      static_for(std::make_integer_sequence<int, PF - 1>{}, step);   // first fragment reads go out behind the bias reads
      if constexpr (PIPE) lds_wait4<PF - 1>(b0, b1, b2, b3);
      const u32x4_t bq[4] = {b0, b1, b2, b3};
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc0[4 * g + e] = acc1[4 * g + e] = __uint_as_float(bq[g][e]);
    } else {
      static_for(std::make_integer_sequence<int, PF - 1>{}, step);
    }
    static_for(std::make_integer_sequence<int, NR>{}, [&](auto s_c) {
      step(std::integral_constant<int, decltype(s_c)::value + PF - 1>{});
    });

    stamp(1);
    // ---- fused epilogue.  Register i <-> channel nb + (i&3) + 8*(i>>2) + 4*h of pixel (row, col).
    if (EPI == EPI_POOL_H2) {
      const int Ho = H >> 1, to = t0 >> 1;
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = (EARLY_RELU ? acc0[i] : relu1(acc0[i], rlim)) + relu1(acc1[i], rlim);   // 1/2 is in the weights
      T* o = (T*)a.out + (((size_t)b * Ho + to) * W + col) * COUT + nb;
      const bool ok = (to < Ho) && col_ok;
      if (sizeof(T) == 4) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (ok) *(float4*)((float*)o + 8 * g + 4 * h) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
      } else {
#pragma unroll
        for (int g = 0; g < 4; g += 2) {  // groups (g, g+1): swap halves so each lane owns 8 consecutive channels
          const unsigned a0 = pack_bf16x2(v[4 * g], v[4 * g + 1]), a1 = pack_bf16x2(v[4 * g + 2], v[4 * g + 3]);
          const unsigned b0 = pack_bf16x2(v[4 * g + 4], v[4 * g + 5]), b1 = pack_bf16x2(v[4 * g + 6], v[4 * g + 7]);
          const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
          // lanes < 32: [own g | upper's g] = channels 8g..8g+7; lanes >= 32: [lower's g+1 | own g+1] = 8g+8..8g+15
          if (ok) *(uint4*)((bf16_t*)o + 8 * g + 8 * h) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
      }
    } else if (EPI == EPI_POOL_2X2) {
      const int Ho = H >> 1, Wo = W >> 1, to = t0 >> 1;
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float s = (EARLY_RELU ? acc0[i] : relu1(acc0[i], rlim)) + relu1(acc1[i], rlim);      // 1/4 is in the weights
        v[i] = s + __shfl_xor(s, 1, 64);                                  // + the neighbouring column (lane r ^ 1)
      }
      const int fo = col >> 1;
      if (to < Ho && fo < Wo && (r & 1) == 0) {
        T* o = (T*)a.out + (((size_t)b * Ho + to) * Wo + fo) * COUT + nb;
        if (sizeof(T) == 4) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *(float4*)((float*)o + 8 * g + 4 * h) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
        } else {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *(uint2*)((bf16_t*)o + 8 * g + 4 * h) =
                make_uint2(pack_bf16x2(v[4 * g], v[4 * g + 1]), pack_bf16x2(v[4 * g + 2], v[4 * g + 3]));
        }
      }
    } else if (EPI == EPI_MEAN_T) {
      if (t0 + 1 < H) {          // wave-uniform: only the last row pair of an odd H takes the other branch
#pragma unroll
        for (int i = 0; i < 16; ++i) cs[i] += (EARLY_RELU ? acc0[i] : relu1(acc0[i], rlim)) + relu1(acc1[i], rlim);
      } else if (t0 < H) {
#pragma unroll
        for (int i = 0; i < 16; ++i) cs[i] += (EARLY_RELU ? acc0[i] : relu1(acc0[i], rlim));
      }
    } else if (EPI == EPI_RAW) {
      float* o0 = a.raw_out + (((size_t)b * H + t0) * W + col) * COUT + nb + 4 * h;
      float* o1 = o0 + (size_t)W * COUT;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (col_ok && t0 < H)
          *(float4*)(o0 + 8 * g) = make_float4(acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]);
        if (col_ok && t0 + 1 < H)
          *(float4*)(o1 + 8 * g) = make_float4(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]);
      }
    } else {  // EPI_PLAIN
      const bool r0ok = col_ok && t0 < H, r1ok = col_ok && t0 + 1 < H;
      float v0[16], v1[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {     // (the statistics forms are pre-BatchNorm outputs: never a ReLU, launcher-checked)
        v0[i] = (!STATS && a.relu) ? relu1(acc0[i], rlim) : acc0[i];
        v1[i] = (!STATS && a.relu) ? relu1(acc1[i], rlim) : acc1[i];
      }
      if (STATS) {
        // Per-lane sums over the rows this lane's COLUMN walks; whether the column counts is a property of the lane, applied
        // once in front of the cross-lane reduction below (a select, so whatever a masked lane accumulated -- its operands
        // are staged zeros here, but were stray LDS bytes in the 30-column experiment -- cannot reach a sum).  Rows past
        // the image are wave-uniform.  4 VALU operations per element pair instead of 7: this epilogue was as long as the
        // unit's 18 MFMAs.
        if (t0 + 1 < H) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            st1[i] += v0[i] + v1[i];
            st2[i] = fmaf(v0[i], v0[i], fmaf(v1[i], v1[i], st2[i]));
          }
        } else if (t0 < H) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            st1[i] += v0[i];
            st2[i] = fmaf(v0[i], v0[i], st2[i]);
          }
        }
      }
      T* o0 = (T*)a.out + (((size_t)b * H + t0) * W + col) * COUT + nb;
      T* o1 = o0 + (size_t)W * COUT;
      if (sizeof(T) == 4) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (r0ok) *(float4*)((float*)o0 + 8 * g + 4 * h) = make_float4(v0[4 * g], v0[4 * g + 1], v0[4 * g + 2], v0[4 * g + 3]);
          if (r1ok) *(float4*)((float*)o1 + 8 * g + 4 * h) = make_float4(v1[4 * g], v1[4 * g + 1], v1[4 * g + 2], v1[4 * g + 3]);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
          unsigned a0 = pack_bf16x2(v0[4 * g], v0[4 * g + 1]), a1 = pack_bf16x2(v0[4 * g + 2], v0[4 * g + 3]);
          unsigned b0 = pack_bf16x2(v0[4 * g + 4], v0[4 * g + 5]), b1 = pack_bf16x2(v0[4 * g + 6], v0[4 * g + 7]);
          auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
          auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
          if (r0ok) *(uint4*)((bf16_t*)o0 + 8 * g + 8 * h) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          a0 = pack_bf16x2(v1[4 * g], v1[4 * g + 1]); a1 = pack_bf16x2(v1[4 * g + 2], v1[4 * g + 3]);
          b0 = pack_bf16x2(v1[4 * g + 4], v1[4 * g + 5]); b1 = pack_bf16x2(v1[4 * g + 6], v1[4 * g + 7]);
          s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
          s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
          if (r1ok) *(uint4*)((bf16_t*)o1 + 8 * g + 8 * h) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
        }
      }
    }
  };

  // one iteration at ring phase PH: prefetch block it+2, run this wave's units, publish the prefetched block
  auto iteration = [&](auto ph_c, int it) {
    constexpr int PH = decltype(ph_c)::value;
    const bool pf = (it + 1 < niter);
    if (pf) {
      if (DMA) stage_dma(it + 2, (PH + 2) % 3); else stage_load(it + 2);
    }
    stamp(0);
#pragma unroll
    for (int uu = 0; uu < C::UPW; ++uu) {
      if (MG == 1) {
        if (uu == 0) unit(ph_c, std::integral_constant<int, 0>{}, it);
        if (uu == 1) unit(ph_c, std::integral_constant<int, 1>{}, it);
      } else {  // MG == 2: row pair uu*2 + mg, dispatched on the (wave-uniform) M group
        if (mg == 0) {
          if (uu == 0) unit(ph_c, std::integral_constant<int, 0>{}, it);
          if (uu == 1) unit(ph_c, std::integral_constant<int, 2>{}, it);
        } else {
          if (uu == 0) unit(ph_c, std::integral_constant<int, 1>{}, it);
          if (uu == 1) unit(ph_c, std::integral_constant<int, 3>{}, it);
        }
      }
    }
    stamp(2);
    if (pf && !DMA) stage_store((PH + 2) % 3);
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA pieces of block it+2 have landed
    stamp(3);
    __syncthreads();
    stamp(4);
  };
  static_assert(C::UPW <= 2 && MG <= 2, "unit dispatch above covers UPW <= 2, MG <= 2");

  stamp(5);   // prologue
  for (int it = 0; it < niter; it += 3) {
    iteration(std::integral_constant<int, 0>{}, it);
    if (it + 1 < niter) iteration(std::integral_constant<int, 1>{}, it + 1);
    if (it + 2 < niter) iteration(std::integral_constant<int, 2>{}, it + 2);
  }

#ifdef DFA_STAMPS
  if (lane == 0 && blockIdx.x < 2048 && (EPI == EPI_MEAN_T || STATS)) {
    long long* d = g_diag + ((size_t)blockIdx.x * 4 + (wave & 3)) * 8;
    for (int k = 0; k < 6; ++k) d[k] = seg[k];
    d[6] = t_begin;
    d[7] = __builtin_amdgcn_s_memtime();
  }
#endif
  if (STATS && a.stats_partial) {
    // per-channel sums: reduce over the 32 pixel lanes of each half-wave, then over the M groups through LDS
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      st1[i] = col_ok ? st1[i] : 0.f;       // lanes whose column lies outside the image (or the strip) do not count
      st2[i] = col_ok ? st2[i] : 0.f;
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
        st1[i] += __shfl_xor(st1[i], off, 64);
        st2[i] += __shfl_xor(st2[i], off, 64);
      }
    }
    float* red = (float*)(smem + C::RING_BYTES + C::BIAS_BYTES);
    if (r == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = (i & 3) + 8 * (i >> 2) + 4 * h;
        red[((mg * NSL + nsl) * 32 + c) * 2] = st1[i];
        red[((mg * NSL + nsl) * 32 + c) * 2 + 1] = st2[i];
      }
    }
    __syncthreads();
    if (tid < NSL * 32) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int g = 0; g < MG; ++g) { s1 += red[(g * NSL * 32 + tid) * 2]; s2 += red[(g * NSL * 32 + tid) * 2 + 1]; }
      float* dst = a.stats_partial + ((size_t)(blockIdx.x * gridDim.y + blockIdx.y) * (NSL * 32) + tid) * 2;
      dst[0] = s1;
      dst[1] = s2;
    }
  }
  if (EPI == EPI_MEAN_T) {
    // embedding rows: for each channel the 32 lanes of a half-wave hold 32 consecutive feature columns
    if (col_ok) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = nb + (i & 3) + 8 * (i >> 2) + 4 * h;
        a.emb[((size_t)b * COUT + c) * W + col] = cs[i] * a.inv_h;
      }
    }
  }
}

// host-side launcher
template <typename T, int CIN, int NSL, int MG, int RP, int MT, int EPI, int MINW, bool ACCIN = false, bool DMA = false,
          bool STATS = false, int PFD = -1>
hipError_t launch_conv3x3(const ConvArgs& a0, hipStream_t stream) {
  using C = ConvCfg<T, CIN, NSL, MG, RP, EPI>;
  ConvArgs a = a0;
  a.nstrips = (a.W + 31) / 32;
  if (STATS && a.relu) return hipErrorInvalidValue;     // the statistics forms store pre-BatchNorm outputs
  auto kern = conv3x3_mfma_kernel<T, CIN, NSL, MG, RP, MT, EPI, MINW, ACCIN, DMA, STATS, PFD>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(a.B * a.nstrips, a.COUT / (NSL * 32), 1), block(C::NT, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, C::LDS_BYTES, stream, a);
#ifdef DFA_STAMPS
  if ((EPI == EPI_MEAN_T || STATS) && sizeof(T) == 2) {
    static int calls = 0;
    if (++calls == 30) {
      static long long hbuf[2048 * 4 * 8];
      hipDeviceSynchronize();
      hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(g_diag), sizeof(hbuf));
      const int nw = (grid.x < 2048 ? grid.x : 2048) * 4;
      double m[8] = {0}; long long tmin = hbuf[6], tmax = hbuf[7];
      for (int i = 0; i < nw; ++i) { for (int k = 0; k < 6; ++k) m[k] += hbuf[i * 8 + k]; m[6] += hbuf[i * 8 + 7] - hbuf[i * 8 + 6];
        if (hbuf[i * 8 + 6] < tmin) tmin = hbuf[i * 8 + 6]; if (hbuf[i * 8 + 7] > tmax) tmax = hbuf[i * 8 + 7]; }
      fprintf(stderr, "[stamps conv3x3<cin %d, epi %d, stats %d>] waves %d  mean cycles/wave: stage_issue %.0f  mfma_loop %.0f  epilogue %.0f  vmcnt_wait %.0f  barrier %.0f  prologue %.0f  lifetime %.0f  kernel span %lld\n",
              CIN, EPI, (int)STATS, nw, m[0] / nw, m[1] / nw, m[2] / nw, m[3] / nw, m[4] / nw, m[5] / nw, m[6] / nw, tmax - tmin);
    }
  }
#endif
  return hipGetLastError();
}

}  // namespace dfa
