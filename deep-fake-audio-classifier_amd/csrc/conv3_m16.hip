// conv3_m16.hip -- CNN2D block 3 (Conv2d(64,128,3,p=1)+BN+ReLU, mean over T; src/model.py:27-29,37) in bf16 on
// v_mfma_f32_16x16x32_bf16.  Same decomposition as conv3x3_mfma.h (workgroup = utterance x 32-column strip x 128
// channels walking down T over a 3-block LDS ring, the wave's 9 x 64 x 32 weight slice resident in 144 VGPRs, LDS-DMA
// staging, asm-pipelined fragment reads), re-tiled for the 16x16 shape: under a dense MFMA stream MI355X holds a higher
// clock on the 16x16x32 form than on 32x32x16 at the same FLOPs per cycle (MI355X_MICROARCH.md, DVFS note 7), so the
// same work finishes sooner.
//
// Tiling.  A operand = weights (16 channels x 32 input channels), B operand = activations (32 input channels x 16
// pixels); D: lane (p = lane&15, q = lane>>4) holds pixel p, channels 4q..4q+3.  A wave owns 32 channels (2 A tiles) x
// 32 pixels (2 B tiles) x 2 rows = 8 accumulators of 4 registers.  One ds_read_b128 fetches, for 16 pixels, the 8 input
// channels 32*kk + 8q.. of one tap; it feeds up to 4 MFMAs (2 channel tiles x the two rows sharing that input row) --
// the same LDS-read-to-MFMA-cycle ratio as the 32x32x16 kernel.
//
// LDS image: pixel slot s owns 128 bytes (64 channels); its 16-byte chunk c sits at physical chunk c ^ (s & 6).  With
// the ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27}, ...) this is conflict-free for all three tap columns, both
// pixel tiles and both k-groups (exhaustive check in tests/test_host_api.py::test_m16_swizzle_is_conflict_free).
#include "dfa_internal.h"
#include <algorithm>
#include <stdio.h>
#include <vector>

namespace dfa {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

namespace m16 {
// A workgroup OWNS SW = 30 output columns (180 = 6 x 30) and loads the SP = 32 columns f0-1 .. f0+30 around them: a ring block is
// then exactly two 1-KiB LDS-DMA pieces per wave (no partial piece, no branch in the staging code, which runs inside the unit's
// asm-read window).  The two 16-pixel MFMA tiles still cover 32 columns; the last two belong to the next strip and are
// dropped (their inputs, slots 32 / 33, are whatever follows in LDS -- MFMA columns are independent).
constexpr int PB = 128, CPP = 8, SP = 32, SW = 30, ROWB = SP * PB, BR = 2, NSL = 4, NT = 256;
constexpr int NCH = BR * SP * CPP, NLD = (NCH + NT - 1) / NT;
constexpr int RING_BYTES = 3 * BR * ROWB, BIAS_BYTES = NSL * 32 * 4, TOT_BYTES = NT * 64;   // running time-mean total: 16 floats per lane
constexpr int LDS_BYTES = RING_BYTES + BIAS_BYTES + TOT_BYTES;
__device__ __forceinline__ int swz(int slot) { return slot & 6; }
}  // namespace m16

__device__ __forceinline__ f32x4_t mma16(const uint4& w, const uint4& x, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), c, 0, 0, 0);
}

// PIPE = false: the compiler-scheduled twin (same arithmetic; the GPU tests require bit-identical output)
// TRAIN = true: the train-mode forward of the same layer (src/train.py:71): the pre-BatchNorm output z is stored (bf16)
// and the per-channel sum / sum of squares of the stored values ride along for the batch statistics (per-workgroup
// partials in conv3x3_mfma's STATS layout); the weights come unfolded, ReLU and the time mean are separate passes.
#ifdef DFA_STAMPS   // diagnostic build (make stamps): shader-clock and real-time stamps around the main loop -> the clock the chip holds
static __device__ long long g_diag16[4096 * 2];
#endif

template <bool PIPE, bool TRAIN = false>
__global__ __launch_bounds__(256, 2) void conv3_m16_meant_kernel(ConvArgs a) {
  using namespace m16;
  constexpr int PF = 4;                  // fragment reads in flight (train form: 2 -> 4 was worth 2 % once the file was built without SLP)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsl = wave;
  const int p = lane & 15, q = lane >> 4;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, xi = bid >> 3;
  const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
  const int b = logical / a.nstrips, strip = logical - b * a.nstrips;
  const int f0 = strip * SW;
  const int H = a.H, W = a.W, COUT = a.COUT;
  const int cout_base = blockIdx.y * (NSL * 32);
  const char* in_b = (const char*)a.in + (size_t)b * H * W * PB;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const float rlim = relu_limit();

  // ---- weights [tap][kk][ca]: 36 fragments = 144 VGPRs for the kernel's lifetime
  uint4 w[9][2][2];
  {
    const uint4* wp = a.wpack + (size_t)(blockIdx.y * NSL + nsl) * 9 * 2 * 2 * 64 + lane;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int ca = 0; ca < 2; ++ca) w[tap][kk][ca] = wp[((tap * 2 + kk) * 2 + ca) * 64];
  }
  float* bias_lds = (float*)(smem + RING_BYTES);
  if (tid < NSL * 32) bias_lds[tid] = a.bias[cout_base + tid];

  // per-lane fragment offsets inside a ring row: slot = p + dx (+16 for the second pixel tile = +2048 bytes, swizzle
  // unchanged), logical chunk 4*kk + q -> kk is one XOR with 64
  int xa[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int slot = p + dx;
    xa[dx] = slot * PB + ((q ^ swz(slot)) << 4);
  }

  // ---- LDS-DMA staging of a ring block (as conv3x3_mfma.h): thread's k-th PHYSICAL chunk, swizzle in the source address
  int s_off[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int g = k * NT + tid;
    const int rowi = g / (SP * CPP), rem = g - rowi * (SP * CPP);
    const int slot = rem / CPP, cph = rem % CPP;
    const int c = cph ^ swz(slot);
    const int f = f0 - 1 + slot;
    const bool ok = (f >= 0) && (f < W);
    s_off[k] = ok ? (rowi * W + f) * PB + c * 16 : -1;
  }
  auto stage_dma = [&](int j, int ringblk) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int g = k * NT + tid;
      static_assert(NCH == NLD * NT, "a ring block is exactly NLD 1-KiB pieces per wave: no conditional piece");
      {
        const int t = BR * j - 1 + g / (SP * CPP);
        const char* src = (s_off[k] >= 0 && t >= 0 && t < H) ? in_b + (ptrdiff_t)(BR * j - 1) * W * PB + s_off[k]
                                                             : (const char*)a.zero_page;
        char* dst = smem + ringblk * BR * ROWB + (k * NT + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };

  f32x4_t cs[2][2];     // eval: running column sums [channel tile][pixel tile]; TRAIN: [0][ca] = sum, [1][ca] = sum of squares
#pragma unroll
  for (int ca = 0; ca < 2; ++ca)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) cs[ca][pb] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int niter_all = (H + BR - 1) / BR;
  // small batches: blockIdx.z walks its own segment [it0, niter) of the time axis (it0 a multiple of the ring period)
  const int it0 = a.seg_iters ? (int)blockIdx.z * a.seg_iters : 0;
  const int niter = a.seg_iters ? min(niter_all, it0 + a.seg_iters) : niter_all;
  stage_dma(it0, 0);
  stage_dma(it0 + 1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  auto unit = [&](auto ph_c, int it) {
    constexpr int PH = decltype(ph_c)::value;
    const int t0 = BR * it;
    f32x4_t acc0[2][2], acc1[2][2];
    constexpr int NR = 4 * 3 * 2 * 2;   // fragment reads in (row i, dx, kk, pb) order
    constexpr int C_RELU0 = 36 + 3;     // acc0's last MFMAs belong to read 35
    constexpr int S_BAR = 4;
    u32x4_t xbuf[PF];
    auto step = [&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      if constexpr (s < NR) {
        constexpr int i = s / 12, dx = (s / 4) % 3, kk = (s / 2) % 2, pb = s % 2;
        constexpr int ringrow = (BR * PH + i) % (3 * BR);
        xbuf[s % PF] = lds_frag<ringrow * ROWB + pb * 16 * PB, PIPE>(lds0 + (xa[dx] ^ (kk << 6)));
        if constexpr (s == S_BAR) {
          // The iteration's barrier stands here, behind the unit's first fragment reads (rows of ring block `it`, published
          // two barriers ago), so the pipeline fill overlaps the wait for the slower waves.  Behind it: the LDS-DMA of block
          // it+2 (overwrites the block the previous unit read) and, from read 24 on, the rows of block it+1 (DMA'd during the
          // previous unit: every wave waits for its own pieces, vmcnt(0), before the barrier).
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if constexpr (PIPE) asm volatile("s_barrier" ::: "memory");
          else __syncthreads();
          if (it + 1 < niter) stage_dma(it + 2, (PH + 2) % 3);
        }
      }
      if constexpr (s >= PF - 1) {
        constexpr int c = s - (PF - 1);
        constexpr int i = c / 12, dx = (c / 4) % 3, kk = (c / 2) % 2, pb = c % 2;
        constexpr int young = (NR - 1 - c) < (PF - 1) ? (NR - 1 - c) : (PF - 1);
        if constexpr (PIPE) lds_wait<young>(xbuf[c % PF]);
        const uint4 xv = __builtin_bit_cast(uint4, xbuf[c % PF]);
#pragma unroll
        for (int ca = 0; ca < 2; ++ca) {
          if constexpr (i <= 2) acc0[ca][pb] = mma16(w[i * 3 + dx][kk][ca], xv, acc0[ca][pb]);
          if constexpr (i >= 1) acc1[ca][pb] = mma16(w[(i - 1) * 3 + dx][kk][ca], xv, acc1[ca][pb]);
        }
        if constexpr (!TRAIN && c == C_RELU0) {   // rows 0..2 done for acc0: its ReLU hides under acc1's last MFMAs
#pragma unroll
          for (int ca = 0; ca < 2; ++ca)
#pragma unroll
            for (int pb2 = 0; pb2 < 2; ++pb2)
#pragma unroll
              for (int e = 0; e < 4; ++e) acc0[ca][pb2][e] = relu1(acc0[ca][pb2][e], rlim);
        }
      }
    };
    {   // bias = accumulator init: channels 16*ca + 4*q + e of this wave's slice
      const unsigned ba = lds0 + RING_BYTES + (nsl * 32 + 4 * q) * 4;
      u32x4_t b0 = lds_frag<0, PIPE>(ba), b1 = lds_frag<64, PIPE>(ba);
      static_for(std::make_integer_sequence<int, PF - 1>{}, step);
      if constexpr (PIPE) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(b0), "+v"(b1) : "n"(PF - 1));
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        acc0[0][pb] = acc1[0][pb] = __builtin_bit_cast(f32x4_t, b0);
        acc0[1][pb] = acc1[1][pb] = __builtin_bit_cast(f32x4_t, b1);
      }
    }
    static_for(std::make_integer_sequence<int, NR>{}, [&](auto s_c) {
      step(std::integral_constant<int, decltype(s_c)::value + PF - 1>{});
    });
    if constexpr (TRAIN) {
      const bool r0 = t0 < H, r1 = t0 + 1 < H;                    // wave-uniform
      // statistics of the fp32 accumulators (as the 32x32x16 kernel); rows / columns outside the image do not count
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        const bool cok = 16 * pb + p < SW && f0 + 16 * pb + p < W;
#pragma unroll
        for (int ca = 0; ca < 2; ++ca) {
          if (cok && r0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { cs[0][ca][e] += acc0[ca][pb][e]; cs[1][ca][e] = fmaf(acc0[ca][pb][e], acc0[ca][pb][e], cs[1][ca][e]); }
          }
          if (cok && r1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { cs[0][ca][e] += acc1[ca][pb][e]; cs[1][ca][e] = fmaf(acc1[ca][pb][e], acc1[ca][pb][e], cs[1][ca][e]); }
          }
        }
      }
      // z stores: v_permlane16_swap between the two pixel tiles hands every lane 8 consecutive channels of ONE pixel --
      // quarter-wave rows q = 0 / 2 keep tile 0 (channels 8*(q/2) .. +7 of the 16-channel tile), rows 1 / 3 take tile 1 --
      // so z leaves as 16-byte stores (4 per lane and unit instead of 8 of 8 bytes)
      const int tile = q & 1;
      const int col = f0 + 16 * tile + p;
      bf16_t* zb = (bf16_t*)a.out + (((size_t)b * H + t0) * W + col) * COUT + cout_base + nsl * 32 + 8 * (q >> 1);
#pragma unroll
      for (int ca = 0; ca < 2; ++ca) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const f32x4_t v0 = r ? acc1[ca][0] : acc0[ca][0], v1 = r ? acc1[ca][1] : acc0[ca][1];
          const auto d0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[0], v0[1]), pack_bf16x2(v1[0], v1[1]), false, false);
          const auto d1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(v0[2], v0[3]), pack_bf16x2(v1[2], v1[3]), false, false);
          if (16 * tile + p < SW && col < W && (r ? r1 : r0)) *(uint4*)(zb + (size_t)r * W * COUT + 16 * ca) = make_uint4(d0[0], d1[0], d0[1], d1[1]);
        }
      }
    } else
    if (t0 + 1 < H) {   // wave-uniform
#pragma unroll
      for (int ca = 0; ca < 2; ++ca)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
          for (int e = 0; e < 4; ++e) cs[ca][pb][e] += acc0[ca][pb][e] + relu1(acc1[ca][pb][e], rlim);
    } else if (t0 < H) {
#pragma unroll
      for (int ca = 0; ca < 2; ++ca)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
          for (int e = 0; e < 4; ++e) cs[ca][pb][e] += acc0[ca][pb][e];
    }
  };

  auto iteration = [&](auto ph_c, int it) { unit(ph_c, it); };      // (DMA issue, DMA wait and the barrier are inside the unit)
  // eval: total over the canonical chunks.  Element k of thread tid lives at [k][tid] (16-byte stride between lanes): the
  // [tid][k] order of round 2 put lanes 64 B apart, a 4-way bank conflict on every ds_read/write_b128 -- the source of block 3's
  // SQ_LDS_BANK_CONFLICT (4.4 M of 55 M LDS cycles, profiles/r02_sq_summary.csv; the fragment reads are conflict-free).
  f32x4_t* const tot = (f32x4_t*)(smem + RING_BYTES + BIAS_BYTES) + tid;
  if constexpr (!TRAIN) {
#pragma unroll
    for (int k = 0; k < 4; ++k) tot[k * NT] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  // outer loop over the canonical chunks of the time mean (chunk_iters iterations, a multiple of 6; one chunk = the whole
  // walk when unset), inner loop = the ring walk itself, unchanged; a chunk's sum is flushed once, outside the hot loop
  const int chunk = (!TRAIN && a.chunk_iters > 0) ? a.chunk_iters : niter_all + 3;
#ifdef DFA_STAMPS
  const long long st_c = __builtin_amdgcn_s_memtime(), st_r = __builtin_amdgcn_s_memrealtime();
#else
  long long st_c = 0, st_r = 0;          // production build: the runtime held-clock probe (ConvArgs::clock_stamps)
  const bool probe = !TRAIN && a.clock_stamps != nullptr;
  if (probe) {
    st_c = __builtin_amdgcn_s_memtime();
    st_r = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): no scalar-memory return may be outstanding inside the counted LDS pipeline
  }
#endif
  for (int c0 = it0; c0 < niter; c0 += chunk) {
    const int cend = min(niter, c0 + chunk);
    for (int it = c0; it < cend; it += 3) {
      iteration(std::integral_constant<int, 0>{}, it);
      if (it + 1 < cend) iteration(std::integral_constant<int, 1>{}, it + 1);
      if (it + 2 < cend) iteration(std::integral_constant<int, 2>{}, it + 2);
    }
    if constexpr (!TRAIN) {
      if (a.seg_iters) {                                 // split: the chunk sum (unscaled) goes to its own slab
        float* e0 = a.emb + (size_t)(c0 / chunk) * a.emb_seg_stride;
#pragma unroll
        for (int ca = 0; ca < 2; ++ca)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) {
            const int col = f0 + 16 * pb + p;
            if (16 * pb + p < SW && col < W) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                e0[((size_t)b * COUT + cout_base + nsl * 32 + 16 * ca + 4 * q + e) * W + col] = cs[ca][pb][e];
            }
          }
      } else {
#pragma unroll
        for (int ca = 0; ca < 2; ++ca)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) tot[(ca * 2 + pb) * NT] += cs[ca][pb];
      }
#pragma unroll
      for (int ca = 0; ca < 2; ++ca)
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) cs[ca][pb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  }
#ifdef DFA_STAMPS
  if (tid == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 4096) {
    g_diag16[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c;
    g_diag16[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r;
  }
#else
  if (probe && tid == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 1024) {
    a.clock_stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - st_c;
    a.clock_stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - st_r;
  }
#endif

  if constexpr (TRAIN) {
    // per-channel sums over this workgroup's pixels: the 16 pixel lanes of a quarter-wave hold the same 8 channels
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int ca = 0; ca < 2; ++ca)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = cs[k][ca][e];
#pragma unroll
          for (int off = 8; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
          cs[k][ca][e] = v;
        }
    if (p == 0 && a.stats_partial) {
      float* dst = a.stats_partial + ((size_t)(blockIdx.x * gridDim.y + blockIdx.y) * (NSL * 32) + nsl * 32 + 4 * q) * 2;
#pragma unroll
      for (int ca = 0; ca < 2; ++ca)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dst[(16 * ca + e) * 2] = cs[0][ca][e];
          dst[(16 * ca + e) * 2 + 1] = cs[1][ca][e];
        }
    }
    return;
  }
  if (a.seg_iters) return;       // split: the chunk sums are out; the classifier kernel adds and scales them
  // embedding rows [b][channel][col]: 16 consecutive columns per (channel, quarter-wave)
#pragma unroll
  for (int ca = 0; ca < 2; ++ca)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const int col = f0 + 16 * pb + p;
      const f32x4_t tv = tot[(ca * 2 + pb) * NT];
      if (16 * pb + p < SW && col < W) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = cout_base + nsl * 32 + 16 * ca + 4 * q + e;
          a.emb[((size_t)b * COUT + c) * W + col] = tv[e] * a.inv_h;
        }
      }
    }
}

// w[COUT][64][3][3] (+ folded eval BatchNorm) -> wpack16[COUT/32][9 taps][2 kk][2 ca][64 lanes] x 16 bytes:
// lane (c = lane&15, q = lane>>4), element j = s[co] * w[co = 32*slice + 16*ca + c][ci = 32*kk + 8*q + j][tap]
__global__ void fold_pack_conv3x3_m16_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                             const float* __restrict__ g, const float* __restrict__ beta,
                                             const float* __restrict__ mean, const float* __restrict__ var, int cin,
                                             int cout, uint4* __restrict__ wpack, int fold) {
  const int total = (cout / 32) * 9 * 2 * 2 * 64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  (void)b; (void)beta; (void)mean;
  if (i >= total) return;
  const int lane = i & 63;
  int rest = i >> 6;
  const int ca = rest & 1; rest >>= 1;
  const int kk = rest & 1; rest >>= 1;
  const int tap = rest % 9;
  const int slice = rest / 9;
  const int co = slice * 32 + 16 * ca + (lane & 15), q = lane >> 4;
  const float s = fold ? g[co] / sqrtf(var[co] + kBnEps) : 1.f;   // fold = 0: raw weights (train mode)
  bf16_t v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = float_to_bf16(w[((size_t)co * cin + 32 * kk + 8 * q + j) * 9 + tap] * s);
  wpack[i] = *reinterpret_cast<const uint4*>(v);
}

hipError_t launch_fold_pack_conv3x3_m16(const float* w, const float* b, const float* g, const float* beta,
                                        const float* mean, const float* var, int cin, int cout, uint4* wpack,
                                        hipStream_t s, int fold) {
  const int total = (cout / 32) * 9 * 2 * 2 * 64;
  hipLaunchKernelGGL(fold_pack_conv3x3_m16_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, b, g, beta, mean, var,
                     cin, cout, wpack, fold);
  return hipGetLastError();
}

template <bool PIPE, bool TRAIN>
static hipError_t launch_m16_t(const ConvArgs& a, hipStream_t stream) {
  auto kern = conv3_m16_meant_kernel<PIPE, TRAIN>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, m16::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int nseg = a.seg_iters ? ((a.H + m16::BR - 1) / m16::BR + a.seg_iters - 1) / a.seg_iters : 1;
  hipLaunchKernelGGL(kern, dim3(a.B * a.nstrips, a.COUT / 128, nseg), dim3(256), m16::LDS_BYTES, stream, a);
#ifdef DFA_STAMPS
  {
    static int calls = 0;
    if (++calls == 4000) {          // after seconds of back-to-back launches on random data: the clock has settled
      static long long hbuf[4096 * 2];
      hipDeviceSynchronize();
      hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(g_diag16), sizeof(hbuf));
      const int n = a.B * a.nstrips < 4096 ? a.B * a.nstrips : 4096;
      std::vector<double> ghz;
      double cyc = 0;
      for (int i = 0; i < n; ++i)
        if (hbuf[2 * i + 1] > 0) { ghz.push_back(hbuf[2 * i] / (hbuf[2 * i + 1] * 10.0)); cyc += hbuf[2 * i]; }
      std::sort(ghz.begin(), ghz.end());
      if (!ghz.empty())
        fprintf(stderr, "[stamps m16] workgroups %zu  main loop %.0f shader cycles/wave  in-kernel clock median %.3f GHz (min %.3f max %.3f)\n",
                ghz.size(), cyc / ghz.size(), ghz[ghz.size() / 2], ghz.front(), ghz.back());
    }
  }
#endif
  return hipGetLastError();
}

hipError_t launch_cnn2d_block3_m16(const ConvArgs& a0, hipStream_t stream, int pipe) {
  ConvArgs a = a0;
  a.nstrips = (a.W + m16::SW - 1) / m16::SW;
  return pipe ? launch_m16_t<true, false>(a, stream) : launch_m16_t<false, false>(a, stream);
}

// train-mode forward of block 3: a.out = z [B][H][W][128] bf16, a.stats_partial = [B*nstrips][128][2]
hipError_t launch_train_fwd3_m16(const ConvArgs& a0, hipStream_t stream, int pipe) {
  ConvArgs a = a0;
  a.nstrips = (a.W + m16::SW - 1) / m16::SW;
  return pipe ? launch_m16_t<true, true>(a, stream) : launch_m16_t<false, true>(a, stream);
}

}  // namespace dfa
