// conv_split.hip -- the parity-grade fast mode (DFA_PREC_BF16X3): CNN2D blocks 2 and 3 (src/model.py:21-29,37) with every
// fp32 value carried as a PAIR of bf16 numbers, v = hi + lo (hi = bf16(v), lo = bf16(v - hi): 16 significant bits), and
// every product taken as three bf16 MFMAs accumulated in fp32:
//     w * x  ~=  w_hi*x_hi + w_lo*x_hi + w_hi*x_lo            (the dropped w_lo*x_lo term is 2^-18 of the product)
// Products of bf16 numbers are exact in the fp32 accumulator, so the result differs from the exact-fp32 path
// (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate) only by the 2^-17 representation error of the operands: logits stay
// within 1e-4 of the reference (golden tests) at 3/16 of the fp32-MFMA cost.
//
// Layout.  A split activation pixel is [hi: C bf16][lo: C bf16] (4C bytes, the size of the fp32 pixel), channels-last as
// everywhere else; conv1.hip writes a1 in this form, the block-2 epilogue writes a2 in this form.  In LDS a pixel slot keeps
// that layout; its 16-byte chunk c sits at physical chunk c ^ swz(slot) with swz = slot & 6 (128-byte pixels, the
// conv3_m16.hip swizzle) or (slot & 7) << 1 (256-byte pixels): conflict-free for the gfx950 ds_read_b128 lane groups over
// all tap columns, pixel tiles, k-steps and hi/lo halves (exhaustive check: tests/test_host_api.py).
//
// Tiling (v_mfma_f32_16x16x32_bf16, as conv3_m16.hip): workgroup = (utterance, 32-column strip, ALL output channels)
// walking down T over a 3-block LDS ring of input rows (LDS-DMA staging, each input element leaves HBM once); wave = 16
// output channels x 32 pixels x 2 rows (16 accumulator registers).  The wave's hi and lo weight fragments for all 9 taps
// stay in registers (72 VGPRs for 32 input channels, 144 for 64).  An x_hi fragment read (ds_read_b128: 8 channels of 16
// pixels) feeds 4 MFMAs (w_hi and w_lo, for the two output rows sharing the input row), an x_lo fragment 2.
#include "dfa_internal.h"

namespace dfa {

typedef __attribute__((ext_vector_type(4))) float f32x4_t;

// SPLIT = false: the same kernel on plain bf16 pixels (one MFMA per product) -- the training step's data-gradient
// convolutions (dz3 -> da2 with 128 input channels in ONE launch, dz2 -> da1), EPI_PLAIN_BF16 epilogue.
template <int CIN, bool SPLIT = true>
struct SplitCfg {
  static constexpr int PB = CIN * (SPLIT ? 4 : 2);  // bytes per pixel ([hi CIN bf16][lo CIN bf16] when split)
  static constexpr int CPP = PB / 16;            // 16-byte chunks per pixel (8 or 16); the lo half starts at chunk CPP/2
  static constexpr int KK = CIN / 32;            // k-steps of 32 input channels
  static constexpr int HL = SPLIT ? 2 : 1;
  static_assert(PB == 128 || PB == 256, "swizzles exist for 128- and 256-byte pixels");
  // A workgroup OWNS SW = 30 output columns (180 = 6 x 30) and loads the SP = 32 columns f0-1 .. f0+30 around them (as
  // conv3_m16.hip: a ring block is a whole number of 1-KiB LDS-DMA pieces per wave, the staging code has no branch and can run
  // inside the unit's asm-read window; the two MFMA tiles' last two columns belong to the next strip and are dropped)
  static constexpr int SP = 32, SW = 30, ROWB = SP * PB, BR = 2;
  static constexpr int RING_BYTES = 3 * BR * ROWB;
  static __device__ __forceinline__ int swz(int slot) { return CPP == 8 ? (slot & 6) : ((slot & 7) << 1); }
};

enum { SPLIT_EPI_POOL_H2 = 0, SPLIT_EPI_MEAN_T = 1, SPLIT_EPI_PLAIN_BF16 = 2 };

static __device__ __forceinline__ f32x4_t mma16s(const uint4& w, const uint4& x, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), c, 0, 0, 0);
}

#ifdef DFA_STAMPS   // diagnostic build (make stamps): per-wave cycle split of an iteration, printed by the launcher
static __device__ long long g_diag_split[4096 * 8];
#endif

// NW waves = NW*16 output channels = COUT.  PIPE = false: the compiler-scheduled twin (bit-identical output).
template <int CIN, int NW, int EPI, bool PIPE, bool SPLIT = true>
__global__ __launch_bounds__(64 * NW, 2) void conv_split_kernel(ConvArgs a) {
  using C = SplitCfg<CIN, SPLIT>;
  constexpr int PB = C::PB, CPP = C::CPP, KK = C::KK, HL = C::HL, SP = C::SP, SW = C::SW, ROWB = C::ROWB, BR = C::BR, NT = 64 * NW;
  static_assert(SPLIT || EPI == SPLIT_EPI_PLAIN_BF16, "the plain-bf16 form is the data-gradient convolution");
  constexpr int COUT = 16 * NW;
  constexpr int NCH = BR * SP * CPP, NLD = NCH / NT;
  static_assert(NCH == NLD * NT, "a ring block is exactly NLD 1-KiB pieces per wave: no conditional piece");
  constexpr int PF = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 15, q = lane >> 4;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, xi = bid >> 3;
  const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
  const int b = logical / a.nstrips, strip = logical - b * a.nstrips;
  const int f0 = strip * SW;
  const int H = a.H, W = a.W;
  const char* in_b = (const char*)a.in + (size_t)b * H * W * PB;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const float rlim = relu_limit();

  // ---- weights [tap][kk][hi|lo]: 18*KK fragments for the kernel's lifetime
  uint4 w[9][KK][HL];
  {
    const uint4* wp = a.wpack + (size_t)wave * 9 * KK * HL * 64 + lane;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
#pragma unroll
        for (int hl = 0; hl < HL; ++hl) w[tap][kk][hl] = wp[((tap * KK + kk) * HL + hl) * 64];
  }
  float* bias_lds = (float*)(smem + C::RING_BYTES);
  if (tid < COUT) bias_lds[tid] = a.bias[tid];

  // per-lane fragment offsets inside a ring row: slot = p + dx (second pixel tile: +16 slots, swizzle unchanged);
  // logical chunk = hl*(CPP/2) + 4*kk + q  ->  (hl, kk) is one XOR with a multiple of 64 bytes
  int xa[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int slot = p + dx;
    xa[dx] = slot * PB + ((q ^ C::swz(slot)) << 4);
  }

  // ---- LDS-DMA staging of a ring block: the thread's k-th PHYSICAL chunk; the swizzle lives in the source address
  int s_off[NLD];
#pragma unroll
  for (int k = 0; k < NLD; ++k) {
    const int g = k * NT + tid;
    const int rowi = g / (SP * CPP), rem = g - rowi * (SP * CPP);
    const int slot = rem / CPP, cph = rem % CPP;
    const int c = cph ^ C::swz(slot);
    const int f = f0 - 1 + slot;
    const bool ok = (f >= 0) && (f < W);
    s_off[k] = ok ? (rowi * W + f) * PB + c * 16 : -1;
  }
  auto stage_dma = [&](int j, int ringblk) {
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int g = k * NT + tid;
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

  f32x4_t cs[2];        // MEAN_T: running column sums per pixel tile
  cs[0] = cs[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int niter_all = (H + BR - 1) / BR;
  const int it0 = a.seg_iters ? (int)blockIdx.z * a.seg_iters : 0;     // time-axis split for small batches (ConvArgs)
  const int niter = a.seg_iters ? min(niter_all, it0 + a.seg_iters) : niter_all;
  stage_dma(it0, 0);
  stage_dma(it0 + 1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#ifdef DFA_STAMPS
  long long seg[4] = {0, 0, 0, 0};
  long long t_prev = __builtin_amdgcn_s_memtime();
  const long long t_begin = t_prev, r_begin = __builtin_amdgcn_s_memrealtime();
  auto stamp = [&](int k) { const long long t = __builtin_amdgcn_s_memtime(); seg[k] += t - t_prev; t_prev = t; };
#else
  auto stamp = [&](int) {};
#endif
  // PLAIN_BF16: the two output rows of an iteration wait in registers and are stored at the START of the next iteration, in
  // front of its LDS-DMA loads: the iteration's closing vmcnt(0) (needed for the DMA before the barrier) counts stores too, and
  // with the stores issued right before it every iteration paid a full write round trip
  uint4 pend_o[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
  size_t pend_i[2] = {0, 0};
  bool pend_ok[2] = {false, false};
  auto flush_pending = [&]() {
    if constexpr (EPI == SPLIT_EPI_PLAIN_BF16) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
        if (pend_ok[r]) *(uint4*)((bf16_t*)a.out + pend_i[r]) = pend_o[r];
    }
  };
  auto unit = [&](auto ph_c, int it) {
    constexpr int PH = decltype(ph_c)::value;
    const int t0 = BR * it;
    f32x4_t acc0[2], acc1[2];
    constexpr int PER_ROW = 3 * KK * HL * 2;       // fragment reads per input row, in (dx, kk, hi|lo, pixel tile) order
    constexpr int NR = 4 * PER_ROW;
    constexpr int C_RELU0 = 3 * PER_ROW + 2;       // acc0's last MFMAs belong to consume step 3*PER_ROW - 1
    // The data-gradient form keeps its barrier at the iteration boundary: its deferred output stores are lane-conditional, and
    // exec branches are not allowed inside the asm-read window (tools/check_lds_pipeline.py).
    constexpr int S_BAR = 4;
    constexpr bool INBAR = EPI != SPLIT_EPI_PLAIN_BF16;
    u32x4_t xbuf[PF];
    // The 32 <- 64 data gradient carries block 1's dropout mask (a.drop): its two Philox calls per lane are index-only work, so
    // they stand INSIDE the MFMA stream (a quarter and a half of the way in) instead of in the epilogue, where both waves of a
    // SIMD ran them at the same time with the matrix pipe idle.  Branch-free: a zero threshold yields all-ones masks.
    constexpr bool MASK_IN_STREAM = EPI == SPLIT_EPI_PLAIN_BF16 && CIN == 64 && NW == 2;
    unsigned km[2][4] = {{~0u, ~0u, ~0u, ~0u}, {~0u, ~0u, ~0u, ~0u}};
    uint4 rc[2];
    uint2 rk[2];
    // output element index of this lane's 16-byte store of row t0 + r: the row part is wave-uniform (scalar multiplies), the
    // lane part a constant of the kernel -- written as one product chain it was ~19 quarter-rate vector multiplies per unit
    const size_t lane_oi = (size_t)(f0 + 16 * (q & 1) + p) * COUT + 16 * wave + 8 * (q >> 1);
    const size_t row_oi = ((size_t)b * H + t0) * ((size_t)W * COUT);
    auto out_index = [&](int r) { return row_oi + (size_t)r * ((size_t)W * COUT) + lane_oi; };
    auto mask_begin = [&](int r) { drop_counter(a.drop, out_index(r), rc[r], rk[r]); };
    // one Philox round behind every second fragment read: row 0 from read R0, row 1 from read R1
    constexpr int R0 = 4, R1 = R0 + 2 * kDropRounds + 4;
    static_assert(R1 + 2 * kDropRounds < NR, "the rounds fit into the stream");
    auto step = [&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      if constexpr (s < NR) {
        constexpr int i = s / PER_ROW, dx = (s / (KK * HL * 2)) % 3, kk = (s / (HL * 2)) % KK, hl = (s / 2) % HL, pb = s % 2;
        constexpr int ringrow = (BR * PH + i) % (3 * BR);
        constexpr int c0 = hl * (CPP / 2) + 4 * kk;
        xbuf[s % PF] = lds_frag<ringrow * ROWB + pb * 16 * PB, PIPE>(lds0 + (xa[dx] ^ (c0 << 4)));
        if constexpr (MASK_IN_STREAM) {
          if constexpr (s == R0 - 2) mask_begin(0);
          if constexpr (s >= R0 && s < R0 + 2 * kDropRounds && (s - R0) % 2 == 0) philox_round(rc[0], rk[0]);
          if constexpr (s == R0 + 2 * kDropRounds) drop_keep_from(a.drop, rc[0], km[0]);
          if constexpr (s == R1 - 2) mask_begin(1);
          if constexpr (s >= R1 && s < R1 + 2 * kDropRounds && (s - R1) % 2 == 0) philox_round(rc[1], rk[1]);
          if constexpr (s == R1 + 2 * kDropRounds) drop_keep_from(a.drop, rc[1], km[1]);
        }
        if constexpr (INBAR && s == S_BAR) {
          // the iteration's barrier, behind the unit's first fragment reads (rows of ring block `it`, published two barriers ago:
          // the pipeline fill overlaps the wait for the slower waves); behind it the previous unit's output stores, the LDS-DMA
          // of block it+2 (overwrites the block the previous unit read) and, in the second half of the stream, the rows of block
          // it+1 (every wave waited for its own pieces, vmcnt(0), before the barrier)
          stamp(1);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          stamp(2);
          if constexpr (PIPE) asm volatile("s_barrier" ::: "memory");
          else __syncthreads();
          stamp(3);
          if (it + 1 < niter) stage_dma(it + 2, (PH + 2) % 3);
          stamp(0);
        }
      }
      if constexpr (s >= PF - 1) {
        constexpr int c = s - (PF - 1);
        constexpr int i = c / PER_ROW, dx = (c / (KK * HL * 2)) % 3, kk = (c / (HL * 2)) % KK, hl = (c / 2) % HL, pb = c % 2;
        constexpr int young = (NR - 1 - c) < (PF - 1) ? (NR - 1 - c) : (PF - 1);
        if constexpr (PIPE) lds_wait<young>(xbuf[c % PF]);
        const uint4 xv = __builtin_bit_cast(uint4, xbuf[c % PF]);
        if constexpr (i <= 2) {
          acc0[pb] = mma16s(w[i * 3 + dx][kk][0], xv, acc0[pb]);                        // w_hi * (x_hi | x_lo)
          if constexpr (SPLIT && hl == 0) acc0[pb] = mma16s(w[i * 3 + dx][kk][HL - 1], xv, acc0[pb]); // w_lo * x_hi
        }
        if constexpr (i >= 1) {
          acc1[pb] = mma16s(w[(i - 1) * 3 + dx][kk][0], xv, acc1[pb]);
          if constexpr (SPLIT && hl == 0) acc1[pb] = mma16s(w[(i - 1) * 3 + dx][kk][HL - 1], xv, acc1[pb]);
        }
        if constexpr (c == C_RELU0 && EPI != SPLIT_EPI_PLAIN_BF16) {   // rows 0..2 done for acc0: its ReLU hides under acc1's last MFMAs
#pragma unroll
          for (int pb2 = 0; pb2 < 2; ++pb2)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc0[pb2][e] = relu1(acc0[pb2][e], rlim);
        }
      }
    };
    {   // bias = accumulator init: channels 16*wave + 4*q + e
      const unsigned ba = lds0 + C::RING_BYTES + (wave * 16 + 4 * q) * 4;
      u32x4_t b0 = lds_frag<0, PIPE>(ba);
      static_for(std::make_integer_sequence<int, PF - 1>{}, step);
      if constexpr (PIPE) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(b0) : "n"(PF - 1));
      acc0[0] = acc0[1] = acc1[0] = acc1[1] = __builtin_bit_cast(f32x4_t, b0);
    }
    static_for(std::make_integer_sequence<int, NR>{}, [&](auto s_c) {
      step(std::integral_constant<int, decltype(s_c)::value + PF - 1>{});
    });
    if constexpr (EPI == SPLIT_EPI_MEAN_T) {
      if (t0 + 1 < H) {   // wave-uniform
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
          for (int e = 0; e < 4; ++e) cs[pb][e] += acc0[pb][e] + relu1(acc1[pb][e], rlim);
      } else if (t0 < H) {
#pragma unroll
        for (int pb = 0; pb < 2; ++pb)
#pragma unroll
          for (int e = 0; e < 4; ++e) cs[pb][e] += acc0[pb][e];
      }
    } else if constexpr (EPI == SPLIT_EPI_PLAIN_BF16) {
      // no activation: both rows leave as bf16, 8 consecutive channels per lane after permlane16_swap (see below)
      const int tile = q & 1;
      const int col = f0 + 16 * tile + p;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const f32x4_t* acc = r ? acc1 : acc0;
        const auto d0 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(acc[0][0], acc[0][1]), pack_bf16x2(acc[1][0], acc[1][1]), false, false);
        const auto d1 = __builtin_amdgcn_permlane16_swap(pack_bf16x2(acc[0][2], acc[0][3]), pack_bf16x2(acc[1][2], acc[1][3]), false, false);
        const size_t oi = out_index(r);
        uint4 o = make_uint4(d0[0], d1[0], d0[1], d1[1]);
        pend_ok[r] = t0 + r < H && col < W && 16 * tile + p < SW;
        if constexpr (MASK_IN_STREAM) {
          o.x &= km[r][0]; o.y &= km[r][1]; o.z &= km[r][2]; o.w &= km[r][3];
        } else if (a.drop.thresh != 0 && pend_ok[r]) {       // one Philox call per 16-byte store
          unsigned k4[4];
          drop_keep8(a.drop, oi, k4);
          o.x &= k4[0]; o.y &= k4[1]; o.z &= k4[2]; o.w &= k4[3];
        }
        pend_o[r] = o;
        pend_i[r] = oi;
      }
    } else {
      // AvgPool2d((2,1)) over the row pair (the 1/2 is in the weights), split into hi + lo and stored as two bf16 planes of
      // the output pixel.  permlane16_swap hands every lane 8 consecutive channels of ONE pixel tile, so hi and lo leave
      // as one 16-byte store each: rows (q) 0/2 keep pixel tile 0 (channels 8*(q/2) .. +7), rows 1/3 take pixel tile 1.
      const int Ho = H >> 1, to = t0 >> 1;
      unsigned hi[2][2], lo[2][2];   // [pixel tile][dword]
#pragma unroll
      for (int pb = 0; pb < 2; ++pb) {
        float v[4], h[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc0[pb][e] + relu1(acc1[pb][e], rlim);
          h[e] = bf16_to_float(float_to_bf16(v[e]));
        }
        hi[pb][0] = pack_bf16x2(v[0], v[1]); hi[pb][1] = pack_bf16x2(v[2], v[3]);
        lo[pb][0] = pack_bf16x2(v[0] - h[0], v[1] - h[1]); lo[pb][1] = pack_bf16x2(v[2] - h[2], v[3] - h[3]);
      }
      const auto h0 = __builtin_amdgcn_permlane16_swap(hi[0][0], hi[1][0], false, false);
      const auto h1 = __builtin_amdgcn_permlane16_swap(hi[0][1], hi[1][1], false, false);
      const auto l0 = __builtin_amdgcn_permlane16_swap(lo[0][0], lo[1][0], false, false);
      const auto l1 = __builtin_amdgcn_permlane16_swap(lo[0][1], lo[1][1], false, false);
      // after the swap: element [0] = channels 4*(q&~1).. of this lane's tile, element [1] = the next 4 channels
      const int tile = q & 1, cb = 16 * wave + 8 * (q >> 1);
      const int col = f0 + 16 * tile + p;
      if (to < Ho && col < W && 16 * tile + p < SW) {
        bf16_t* o = (bf16_t*)a.out + (((size_t)b * Ho + to) * W + col) * (2 * COUT) + cb;
        *(uint4*)o = make_uint4(h0[0], h1[0], h0[1], h1[1]);
        *(uint4*)(o + COUT) = make_uint4(l0[0], l1[0], l0[1], l1[1]);
      }
    }
  };

  auto iteration = [&](auto ph_c, int it) {
    if constexpr (EPI != SPLIT_EPI_PLAIN_BF16) {
      unit(ph_c, it);              // DMA wait, barrier and DMA issue are inside the unit
    } else {
      constexpr int PH = decltype(ph_c)::value;
      flush_pending();
      if (it + 1 < niter) stage_dma(it + 2, (PH + 2) % 3);
      stamp(0);
      unit(ph_c, it);
      stamp(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stamp(2);
      __syncthreads();
      stamp(3);
    }
  };
  // MEAN_T: running total over the canonical chunks of the time mean (ConvArgs::chunk_iters), 8 floats per lane in LDS
  constexpr int TOT_STRIDE = NT;
  f32x4_t* const tot = (f32x4_t*)(smem + C::RING_BYTES + COUT * 4) + tid;   // element k at [k][tid]: conflict-free b128 accesses (conv3_m16.hip)
  if constexpr (EPI == SPLIT_EPI_MEAN_T) tot[0] = tot[TOT_STRIDE] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int chunk = (EPI == SPLIT_EPI_MEAN_T && a.chunk_iters > 0) ? a.chunk_iters : niter_all + 3;
  for (int c0 = it0; c0 < niter; c0 += chunk) {          // canonical chunks of the time mean; the inner loop is the ring walk
    const int cend = min(niter, c0 + chunk);
    for (int it = c0; it < cend; it += 3) {
      iteration(std::integral_constant<int, 0>{}, it);
      if (it + 1 < cend) iteration(std::integral_constant<int, 1>{}, it + 1);
      if (it + 2 < cend) iteration(std::integral_constant<int, 2>{}, it + 2);
    }
    if constexpr (EPI == SPLIT_EPI_MEAN_T) {
      if (a.seg_iters) {
        float* e0 = a.emb + (size_t)(c0 / chunk) * a.emb_seg_stride;
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
          const int col = f0 + 16 * pb + p;
          if (col < W && 16 * pb + p < SW) {
#pragma unroll
            for (int e = 0; e < 4; ++e) e0[((size_t)b * COUT + 16 * wave + 4 * q + e) * W + col] = cs[pb][e];
          }
        }
      } else {
        tot[0] += cs[0];
        tot[TOT_STRIDE] += cs[1];
      }
      cs[0] = cs[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
  }

  flush_pending();               // the last iteration's rows
#ifdef DFA_STAMPS
  if (lane == 0 && blockIdx.x < 1024 && blockIdx.z == 0 && wave < 2) {
    long long* dd = g_diag_split + ((size_t)blockIdx.x * 2 + wave) * 8;
    for (int k = 0; k < 4; ++k) dd[k] = seg[k];
    dd[4] = __builtin_amdgcn_s_memtime() - t_begin;
    dd[5] = __builtin_amdgcn_s_memrealtime() - r_begin;
    dd[6] = niter - it0;
  }
#endif
  if constexpr (EPI == SPLIT_EPI_MEAN_T) {
    if (a.seg_iters) return;     // split: the classifier kernel adds and scales the chunk sums
    // embedding rows [b][channel][col]: 16 consecutive columns per (channel, quarter-wave)
#pragma unroll
    for (int pb = 0; pb < 2; ++pb) {
      const int col = f0 + 16 * pb + p;
      const f32x4_t tv = tot[pb * TOT_STRIDE];
      if (col < W && 16 * pb + p < SW) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 16 * wave + 4 * q + e;
          a.emb[((size_t)b * COUT + c) * W + col] = tv[e] * a.inv_h;
        }
      }
    }
  }
}

// w[COUT][CIN][3][3] (+ folded eval BatchNorm, * post_scale) -> wsplit[COUT/16][9 taps][CIN/32][hi|lo][64 lanes] x 16 B:
// lane (c = lane&15, q = lane>>4), element j: v = s[co] * w[co = 16*tile + c][ci = 32*kk + 8*q + j][tap] * post_scale;
// hi = bf16(v), lo = bf16(v - hi).  bias[co] = folded bias * post_scale (fp32).
__global__ void fold_pack_conv3x3_split_kernel(const float* __restrict__ w, const float* __restrict__ b,
                                               const float* __restrict__ g, const float* __restrict__ beta,
                                               const float* __restrict__ mean, const float* __restrict__ var, int cin,
                                               int cout, uint4* __restrict__ wpack, float* __restrict__ bias,
                                               float post_scale) {
  const int kkn = cin / 32;
  const int total = (cout / 16) * 9 * kkn * 2 * 64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cout) {
    const float s = g[i] / sqrtf(var[i] + kBnEps);
    bias[i] = ((b[i] - mean[i]) * s + beta[i]) * post_scale;
  }
  if (i >= total) return;
  const int lane = i & 63;
  int rest = i >> 6;
  const int hl = rest & 1; rest >>= 1;
  const int kk = rest % kkn; rest /= kkn;
  const int tap = rest % 9;
  const int tile = rest / 9;
  const int co = tile * 16 + (lane & 15), q = lane >> 4;
  const float s = g[co] / sqrtf(var[co] + kBnEps);
  bf16_t v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float wv = w[((size_t)co * cin + 32 * kk + 8 * q + j) * 9 + tap] * s * post_scale;
    const bf16_t h = float_to_bf16(wv);
    v[j] = hl ? float_to_bf16(wv - bf16_to_float(h)) : h;
  }
  wpack[i] = *reinterpret_cast<const uint4*>(v);
}

hipError_t launch_fold_pack_conv3x3_split(const float* w, const float* b, const float* g, const float* beta,
                                          const float* mean, const float* var, int cin, int cout, uint4* wpack,
                                          float* bias, float post_scale, hipStream_t s) {
  int total = (cout / 16) * 9 * (cin / 32) * 2 * 64;
  if (total < cout) total = cout;
  hipLaunchKernelGGL(fold_pack_conv3x3_split_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, b, g, beta, mean, var,
                     cin, cout, wpack, bias, post_scale);
  return hipGetLastError();
}

template <int CIN, int NW, int EPI, bool PIPE, bool SPLIT = true>
static hipError_t launch_split_t(const ConvArgs& a, hipStream_t stream) {
  auto kern = conv_split_kernel<CIN, NW, EPI, PIPE, SPLIT>;
  constexpr int LDS = SplitCfg<CIN, SPLIT>::RING_BYTES + NW * 16 * 4 + (EPI == SPLIT_EPI_MEAN_T ? 64 * NW * 32 : 0);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int nseg = a.seg_iters ? ((a.H + 1) / 2 + a.seg_iters - 1) / a.seg_iters : 1;
  hipLaunchKernelGGL(kern, dim3(a.B * a.nstrips, 1, nseg), dim3(64 * NW), LDS, stream, a);
#ifdef DFA_STAMPS
  {
    static int calls = 0;
    if (++calls == 30 && PIPE) {
      static long long hbuf[4096 * 8];
      hipDeviceSynchronize();
      hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(g_diag_split), sizeof(hbuf));
      const int nw = (a.B * a.nstrips < 1024 ? a.B * a.nstrips : 1024) * 2;
      double m[6] = {0, 0, 0, 0, 0, 0}, iters = 0;
      for (int i = 0; i < nw; ++i) { for (int k = 0; k < 6; ++k) m[k] += hbuf[i * 8 + k]; iters += hbuf[i * 8 + 6]; }
      fprintf(stderr, "[stamps conv_split<%d,%d,epi %d,split %d>] cycles per wave-iteration: stores + dma issue %.0f  mfma stream + epilogue %.0f  "
                      "dma wait %.0f  barrier %.0f  | lifetime/iteration %.0f  clock %.3f GHz\n", CIN, NW, EPI, (int)SPLIT,
              m[0] / iters, m[1] / iters, m[2] / iters, m[3] / iters, m[4] / iters, m[4] / (m[5] * 10.0));
    }
  }
#endif
  return hipGetLastError();
}

// block 2: a.in = a1 split [B][H][W][2*32], a.out = a2 split [B][H/2][W][2*64]
hipError_t launch_cnn2d_block2_split(const ConvArgs& a0, hipStream_t stream, int pipe) {
  ConvArgs a = a0;
  a.nstrips = (a.W + 29) / 30;
  return pipe ? launch_split_t<32, 4, SPLIT_EPI_POOL_H2, true>(a, stream) : launch_split_t<32, 4, SPLIT_EPI_POOL_H2, false>(a, stream);
}

// block 3: a.in = a2 split [B][H][W][2*64], a.emb = [B][128][W] fp32 (mean over H)
hipError_t launch_cnn2d_block3_split(const ConvArgs& a0, hipStream_t stream, int pipe) {
  ConvArgs a = a0;
  a.nstrips = (a.W + 29) / 30;
  return pipe ? launch_split_t<64, 8, SPLIT_EPI_MEAN_T, true>(a, stream) : launch_split_t<64, 8, SPLIT_EPI_MEAN_T, false>(a, stream);
}

// ---- training: data-gradient convolutions on the same kernel (plain bf16, one launch each)
// da = conv3x3(dz, W') with W'[ci][co][dy'][dx'] = W[co][ci][2-dy'][2-dx'] (input channels = the forward conv's OUTPUT channels):
// wpack[cin/16][9 taps][cout/32][64 lanes] x 16 B, lane (c = lane&15, q = lane>>4), element j = w[co = 32*kk + 8*q + j][ci = 16*tile + c][8 - tap]
__global__ void pack_conv3x3_dgrad_m16_kernel(const float* __restrict__ w, int cin, int cout, uint4* __restrict__ wpack,
                                              float* __restrict__ bias) {
  const int kkn = cout / 32;
  const int total = (cin / 16) * 9 * kkn * 64;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cin) bias[i] = 0.f;
  if (i >= total) return;
  const int lane = i & 63;
  int rest = i >> 6;
  const int kk = rest % kkn; rest /= kkn;
  const int tap = rest % 9;
  const int tile = rest / 9;
  const int ci = tile * 16 + (lane & 15), q = lane >> 4;
  bf16_t v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = float_to_bf16(w[((size_t)(32 * kk + 8 * q + j) * cin + ci) * 9 + (8 - tap)]);
  wpack[i] = *reinterpret_cast<const uint4*>(v);
}

hipError_t launch_pack_conv3x3_dgrad_m16(const float* w, int cin, int cout, uint4* wpack, float* bias, hipStream_t s) {
  int total = (cin / 16) * 9 * (cout / 32) * 64;
  if (total < cin) total = cin;
  hipLaunchKernelGGL(pack_conv3x3_dgrad_m16_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, cin, cout, wpack, bias);
  return hipGetLastError();
}

// block 3 data gradient: a.in = dz3 [B][H][W][128] bf16, a.out = da2 [B][H][W][64] bf16 (one launch: no fp32 partial sums)
hipError_t launch_train_dgrad3_m16(const ConvArgs& a0, hipStream_t stream, int pipe) {
  ConvArgs a = a0;
  a.nstrips = (a.W + 29) / 30;
  return pipe ? launch_split_t<128, 4, SPLIT_EPI_PLAIN_BF16, true, false>(a, stream)
              : launch_split_t<128, 4, SPLIT_EPI_PLAIN_BF16, false, false>(a, stream);
}

// block 2 data gradient: a.in = dz2 [B][H][W][64] bf16, a.out = da1 [B][H][W][32] bf16
hipError_t launch_train_dgrad2_m16(const ConvArgs& a0, hipStream_t stream, int pipe) {
  ConvArgs a = a0;
  a.nstrips = (a.W + 29) / 30;
  return pipe ? launch_split_t<64, 2, SPLIT_EPI_PLAIN_BF16, true, false>(a, stream)
              : launch_split_t<64, 2, SPLIT_EPI_PLAIN_BF16, false, false>(a, stream);
}

}  // namespace dfa
