// conv12_fused.hip -- CNN2D blocks 1 and 2 in ONE kernel (bf16 mode; fp32 features are rounded to bf16 as they are loaded):
//   Conv2d(1,32,3,p=1)+BN+ReLU+AvgPool(2,1)  ->  Conv2d(32,64,3,p=1)+BN+ReLU+AvgPool(2,1)      (src/model.py:15-25)
// The block-1 activation a1 [B,160,180,32] (1.84 MB per utterance, written and re-read through HBM by the two-kernel
// path: 0.94 GB per step at B = 256) never leaves the chip: the workgroup that consumes a ring block of a1 rows
// produces it, from the raw features, straight into the LDS ring of conv3x3_mfma's block-2 main loop.
//
// Block 1 on the matrix cores.  One v_mfma_f32_32x32x16_bf16 has K = 16 = 4 feature rows x (3 taps + 1 zero): for the
// pooled a1 row q, the B operand of a 32-pixel tile holds x[2q-1 .. 2q+2][f-1 .. f+1] and serves BOTH conv rows of the
// pool pair -- the even row 2q through an A operand with weights on feature rows 0..2, the odd row 2q+1 through one
// with weights on rows 1..3.  The folded fp32 weights are split into bf16 hi + lo parts (w = hi + lo to 2^-17), one MFMA
// each: products of bf16 inputs are exact in the fp32 accumulator, so block 1 keeps fp32-level accuracy on bf16 input.
// Cost: 4 MFMAs per a1 tile against the 36 of the block-2 work that consumes it.
//
// Strips are 30 output columns wide: with the two halo columns a ring row has exactly 32 live slots = ONE block-1 MFMA
// tile per a1 row, one tile per wave per iteration (180 = 6 x 30: the block-2 MFMAs still compute 6 x 32 columns, the
// same count the 32-wide strips of the unfused kernel spend on their ragged last strip).
//
// LDS: [block-2 ring 3 x 4 rows x 36 slots x 64 B][bias2][2 x feature-window tiles 10 rows x 36 slots x 8 B].
// A window entry is {x[f-1], x[f], x[f+1], 0} (4 bf16), i.e. one half of a lane's B operand: the lane reads two of them
// (ds_read_b64 x 2).  Pipeline per iteration `it` (one __syncthreads each, like the unfused kernel):
//   global-load the features of ring block it+3 -> registers;  block-2 MFMA unit on ring blocks it, it+1 with the wave's
//   block-1 tile of ring block it+2 (window buffer it&1 -> ring slot (it+2)%3) and the register -> window buffer
//   (it+3)&1 stores threaded through its MFMA stream, so the block-1 VALU work hides in the MFMA shadow.
#include <stdlib.h>

#include "dfa_internal.h"

namespace dfa {

struct Conv12Args {
  const void* x;            // features (bf16 or fp32: template argument TX), element strides below (any layout)
  long long sxb, sxt, sxf;
  const uint4* c1pack;      // [4][64] A operands: even-hi, even-lo, odd-hi, odd-lo (pack.hip: pack_conv1_mfma_kernel)
  const float* c1bias;      // [32]  0.5 * folded bias
  const uint4* wpack;       // block 2 image [2][9][2][64] (pool factor folded)
  const float* bias;        // [64]
  bf16_t* out;              // a2 [B][H1/2][F][64]
  int B, T, F, H1, nstrips;
  int seg_iters;            // time-axis split for small batches: blockIdx.y walks seg_iters iterations (multiple of 6), 0 = all
};

__device__ __forceinline__ float ld_as_float(const float* p) { return *p; }
__device__ __forceinline__ float ld_as_float(const bf16_t* p) { return bf16_to_float(*p); }

namespace c12 {
constexpr int PB = 64, SP = 36, ROWB = SP * PB, BR = 4, NKG = 2, PF = 4, SW = 30;   // SW: output columns per strip
constexpr int RING_BYTES = 3 * BR * ROWB;
constexpr int BIAS2_OFF = RING_BYTES, XW_OFF = BIAS2_OFF + 64 * 4;
constexpr int XROWS = 10, XW_ROWB = SP * 8, XW_BYTES = XROWS * XW_ROWB;
constexpr int DUMMY_OFF = XW_OFF + 2 * XW_BYTES;   // sink for the window stores of out-of-tile taps (keeps them branch-free)
constexpr int LDS_BYTES = DUMMY_OFF + 16;
constexpr int XCOLS = 34;                 // feature columns per ring block: 32 slots + 1 more each side
constexpr int NX = XROWS * XCOLS;
constexpr int NXLD = (NX + 255) / 256;
}  // namespace c12

#ifdef DFA_STAMPS
static __device__ long long g_diag12[2048 * 4 * 8];
#endif

// PIPE = false is the compiler-scheduled twin (plain LDS loads, same arithmetic): the GPU tests require bit-identical output
template <typename TX, bool PIPE>
__global__ __launch_bounds__(256, 2) void conv12_fused_kernel(Conv12Args a) {
  using namespace c12;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nsl = wave & 1, mg = wave >> 1;
  const int r = lane & 31, h = lane >> 5;

  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7, xi = bid >> 3;
  const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
  const int b = logical / a.nstrips, strip = logical - b * a.nstrips;
  const int f0 = strip * SW;
  const int H = a.H1, W = a.F, T = a.T;
  const int nb = nsl * 32;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const float rlim = relu_limit();

  // ---- register-resident operands: block-2 weight slice (72 VGPRs) and the four block-1 A operands (16 VGPRs)
  uint4 w[9][NKG];
  {
    const uint4* wp = a.wpack + (size_t)nsl * 9 * NKG * 64 + lane;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kg = 0; kg < NKG; ++kg) w[tap][kg] = wp[(tap * NKG + kg) * 64];
  }
  uint4 c1w[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) c1w[k] = a.c1pack[k * 64 + lane];

  float* bias2_lds = (float*)(smem + BIAS2_OFF);
  if (tid < 64) bias2_lds[tid] = a.bias[tid];
  for (int i = tid; i < 2 * XW_BYTES / 8; i += 256) *(uint2*)(smem + XW_OFF + i * 8) = make_uint2(0u, 0u);   // window pads
  f32x16_t bias1;            // block-1 bias as the C operand of the tile's first MFMAs (channel e + 8g + 4h <-> element 4g + e)
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float4 bv = *(const float4*)(a.c1bias + 8 * g + 4 * h);
    bias1[4 * g] = bv.x; bias1[4 * g + 1] = bv.y; bias1[4 * g + 2] = bv.z; bias1[4 * g + 3] = bv.w;
  }

  int xa[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int slot = r + dx, s = lds_swz<PB>(slot);
    xa[dx] = slot * PB + (((h ^ (s & 1)) << 4) | ((s >> 1) << 5));
  }

  // ---- feature staging: element e of a ring block = (local row, column c); x column f0 - 2 + c, x row 8j - 3 + row
  const bool t_fast = (a.sxt == 1);
  int xrow[NXLD], xcol[NXLD];
#pragma unroll
  for (int k = 0; k < NXLD; ++k) {
    const int e = k * 256 + tid;
    xrow[k] = t_fast ? e % XROWS : e / XCOLS;
    xcol[k] = t_fast ? e / XROWS : e % XCOLS;
  }
  const TX* xb = (const TX*)a.x + (long long)b * a.sxb;
  unsigned short xreg[NXLD];   // raw loaded bits; out-of-image elements are zeroed when they are stored, not here: a
  bool xok[NXLD];              // select on the loaded value would make the wave wait out the load latency at issue time
  // per-thread constants of the feature loads: the element's offset for ring block 0 and whether its column exists; a ring
  // block only adds the wave-uniform 8*j*sxt (no 64-bit multiplies in the loop)
  long long xoff[NXLD];
  bool xfok[NXLD];
#pragma unroll
  for (int k = 0; k < NXLD; ++k) {
    const int f = f0 - 2 + xcol[k];
    xfok[k] = (k * 256 + tid < NX) && f >= 0 && f < W;
    xoff[k] = (long long)(xrow[k] - 3) * a.sxt + (long long)(xfok[k] ? f : 0) * a.sxf;
  }
  auto x_load = [&](int j) {
    const long long jo = (long long)(8 * j) * a.sxt;        // wave-uniform
#pragma unroll
    for (int k = 0; k < NXLD; ++k) {
      const int t = 8 * j - 3 + xrow[k];
      xok[k] = xfok[k] && (unsigned)t < (unsigned)T;
      const TX* src = xb + (xok[k] ? xoff[k] + jo : 0);     // clamped address, branch-free
      if constexpr (sizeof(TX) == 2) xreg[k] = *(const unsigned short*)src;            // bf16 features: the bits as they are
      else xreg[k] = cvt_out<bf16_t>(ld_as_float(src)).v;                               // fp32 features: RNE on load
    }
  };
  auto x_store = [&](int buf) {   // element (row, c) is tap e of the windows of slots c - e, e = 0..2
#pragma unroll
    for (int k = 0; k < NXLD; ++k) {
      const int base = XW_OFF + buf * XW_BYTES + xrow[k] * XW_ROWB;
      const unsigned short v = xok[k] ? xreg[k] : (unsigned short)0;
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int s = xcol[k] - e;
        const bool ok = (k * 256 + tid < NX) && s >= 0 && s < 32;
        const int off = ok ? base + s * 8 + e * 2 : DUMMY_OFF;
        if constexpr (PIPE) {   // through asm so that the store's place in the in-order LDS queue is known to the counted waits
          asm volatile("ds_write_b16 %0, %1" : : "v"(lds0 + off), "v"((unsigned)v) : "memory");
        } else {
          *(unsigned short*)(smem + off) = v;
        }
      }
    }
  };

  // ---- block 1: this wave's 32-pixel tile of ring block j = a1 row m = wave, slots 0..31, in four pieces that the
  // block-2 unit threads through its MFMA stream (the prologue runs them back to back).
  const int c1_m = wave;
  const unsigned c1_win = lds0 + XW_OFF + ((2 * c1_m + 2 * h) * SP + r) * 8;       // + (j&1)*XW_BYTES
  const int c1_f = f0 - 1 + r;
  const bool c1_fok = c1_f >= 0 && c1_f < W;
  const int c1_dst = (c1_m * SP + r) * PB;                                         // + ringblk*BR*ROWB
  const int c1_sw = lds_swz<PB>(r);
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  struct C1State { u32x2_t w0, w1; f32x16_t e, o; float v[16]; };
  auto c1_issue = [&](C1State& st, int j) {       // two window reads (asm: they join the counted LDS pipeline)
    const unsigned addr = c1_win + (j & 1) * XW_BYTES;
    if constexpr (PIPE) {
      asm volatile("ds_read_b64 %0, %1" : "=v"(st.w0) : "v"(addr));
      asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(st.w1) : "v"(addr), "n"(XW_ROWB));
    } else {
      st.w0 = *(const u32x2_t*)((const __attribute__((address_space(3))) char*)(size_t)addr);
      st.w1 = *(const u32x2_t*)((const __attribute__((address_space(3))) char*)(size_t)(addr + XW_ROWB));
    }
  };
  auto c1_mfma = [&](C1State& st) {               // the windows have landed (caller's counted wait)
    if constexpr (PIPE) asm volatile("" : "+v"(st.w0), "+v"(st.w1));
    const uint4 xv = make_uint4(st.w0[0], st.w0[1], st.w1[0], st.w1[1]);
    st.e = Mma<bf16_t>::run(c1w[0], xv, bias1);
    st.o = Mma<bf16_t>::run(c1w[2], xv, bias1);
    st.e = Mma<bf16_t>::run(c1w[1], xv, st.e);
    st.o = Mma<bf16_t>::run(c1w[3], xv, st.o);
  };
  auto c1_relu = [&](C1State& st, int j) {        // ReLU + pool add; positions outside the image are block 2's zero padding
    const int q = BR * j - 1 + c1_m;
    const float lim = (q >= 0 && q < H && c1_fok) ? __builtin_inff() : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      st.v[i] = __builtin_amdgcn_fmed3f(st.e[i], 0.f, lim) + __builtin_amdgcn_fmed3f(st.o[i], 0.f, lim);
  };
  auto c1_store = [&](C1State& st, int ringblk) { // bf16 pack, half-wave swap -> two 16-byte chunks into the ring
    char* dst = smem + ringblk * (BR * ROWB) + c1_dst;
#pragma unroll
    for (int g = 0; g < 4; g += 2) {   // lanes < 32 end up with channels 8g..8g+7, lanes >= 32 with 8g+8..8g+15
      const unsigned a0 = pack_bf16x2(st.v[4 * g], st.v[4 * g + 1]), a1 = pack_bf16x2(st.v[4 * g + 2], st.v[4 * g + 3]);
      const unsigned b0 = pack_bf16x2(st.v[4 * g + 4], st.v[4 * g + 5]), b1 = pack_bf16x2(st.v[4 * g + 6], st.v[4 * g + 7]);
      const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      if constexpr (PIPE) {
        const u32x4_t pk = {s0[0], s1[0], s0[1], s1[1]};
        asm volatile("ds_write_b128 %0, %1" : : "v"(lds0 + ringblk * (BR * ROWB) + c1_dst + (((g + h) ^ c1_sw) << 4)), "v"(pk) : "memory");
      } else {
        *(uint4*)(dst + (((g + h) ^ c1_sw) << 4)) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
      }
    }
  };
  auto produce_now = [&](int j, int ringblk) {
    C1State st;
    c1_issue(st, j);
    if constexpr (PIPE) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(st.w0), "+v"(st.w1));
    c1_mfma(st);
    c1_relu(st, j);
    c1_store(st, ringblk);
  };

  const int col = f0 + r;
  const bool col_ok = (r < SW) && col < W;
  const int niter = (H + BR - 1) / BR;
  bf16_t* const obase = a.out + ((size_t)b * (H >> 1) * W + (col_ok ? col : 0)) * 64 + nb + 8 * h;   // + to * W * 64 per unit

#ifdef DFA_STAMPS   // diagnostic build (make stamps): per-wave cycle split, printed by the launcher
  long long seg[6] = {0, 0, 0, 0, 0, 0};
  long long t_prev = __builtin_amdgcn_s_memtime();
  const long long t_begin = t_prev, r_begin = __builtin_amdgcn_s_memrealtime();
  auto stamp = [&](int k) { const long long t = __builtin_amdgcn_s_memtime(); seg[k] += t - t_prev; t_prev = t; };
#else
  auto stamp = [&](int) {};
#endif
  // small batches: blockIdx.y walks its own segment [it0, niter) of the time axis.  it0 is a multiple of 6: of the ring
  // period 3 (ring block j lives in slot j % 3) and of the window-buffer period 2 (windows of block j in buffer j & 1)
  const int it0 = a.seg_iters ? (int)blockIdx.y * a.seg_iters : 0;
  const int niter_seg = a.seg_iters ? min(niter, it0 + a.seg_iters) : niter;
  // ---- prologue: windows of blocks it0, it0+1 -> ring blocks 0, 1; windows of block it0+2
  // the window / ring stores of the pipelined build go out through asm: the compiler does not know they are in flight, so the
  // waits in front of the prologue's barriers are explicit (in the loop the last counted wait of a unit is lgkmcnt(0))
  auto lds_drain = [&]() { if constexpr (PIPE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
  __syncthreads();                 // window pads / bias written
  x_load(it0); x_store(0);
  x_load(it0 + 1); x_store(1);
  lds_drain();
  __syncthreads();
  produce_now(it0, 0);
  produce_now(it0 + 1, 1);
  lds_drain();
  __syncthreads();
  x_load(it0 + 2); x_store(0);
  lds_drain();
  __syncthreads();

  // ---- block 2 unit (conv3x3_mfma.h, <bf16, CIN 32, POOL_H2>, asm-pipelined fragment reads) + the block-1 tile of
  // ring block it+2 and the window stores of block it+3 at fixed points of its MFMA stream
  // The packed pooled outputs of a unit are kept (8 registers) and leave -- half-wave swaps + two 16-byte stores -- from inside
  // the NEXT unit's MFMA stream; the feature loads of ring block it+3 are issued from inside the stream as well: neither
  // stands alone between two MFMA streams any more (per-wave stamps, one wave per SIMD: the epilogue was 536 and the
  // loads 301 of an iteration's 3388 cycles).
  unsigned pq[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  bf16_t* po = obase;
  bool pok = false;
  auto flush_pending = [&]() {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const auto s0 = __builtin_amdgcn_permlane32_swap(pq[4 * g], pq[4 * g + 2], false, false);
      const auto s1 = __builtin_amdgcn_permlane32_swap(pq[4 * g + 1], pq[4 * g + 3], false, false);
      if (pok) *(uint4*)(po + 16 * g) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
    }
  };
  auto unit = [&](auto ph_c, auto rp_c, int it) {
    constexpr int PH = decltype(ph_c)::value, RPI = decltype(rp_c)::value;
    f32x16_t acc0, acc1;
    const int t0 = BR * it + 2 * RPI;
    constexpr int NR = 12 * NKG;
    constexpr int S_RELU0 = 9 * NKG + 2;
    // The workgroup barrier of an iteration stands INSIDE the unit, behind fragment read S_BAR: what the unit reads before it
    // (ring rows written two iterations ago) was published by the previous barrier, and what must not start before it -- the
    // window reads of block it+2 (stored by the previous unit), the ring store at C_STORE (overwrites the block the previous
    // unit read) and the reads of the rows the previous unit stored (second half of the stream) -- comes after it.  The
    // unit's pipeline fill (the latency of its first fragment reads, ~400 cycles per iteration in the stamped build) thus
    // overlaps the wait for the slower waves instead of following it.
    // consume steps carrying block-1 pieces: the window reads are issued behind fragment read S_BAR; by the wait of consume
    // step S_BAR + 1 (all but the 3 youngest reads landed, all of them younger than the window reads) they are in registers
    constexpr int S_BAR = 4;
    constexpr int C_MFMA = 5, C_XLOAD = 7, C_PSTORE = 9, C_RELU = 11, C_STORE = 15, C_XSTORE = 19;
    u32x4_t xbuf[PF];
    C1State c1;
    auto step = [&](auto s_c) {
      constexpr int s = decltype(s_c)::value;
      if constexpr (s < NR) {
        constexpr int i = s / (3 * NKG), dx = (s / NKG) % 3, kg = s % NKG;
        constexpr int ringrow = (BR * PH + 2 * RPI + i) % (3 * BR);
        xbuf[s % PF] = lds_frag<ringrow * ROWB, PIPE>(lds0 + (xa[dx] ^ (kg << 5)));
        if constexpr (s == S_BAR) {
          // (this wave's own ring / window stores of the previous unit are complete: that unit's last counted wait is lgkmcnt(0))
          if constexpr (PIPE) asm volatile("s_barrier" ::: "memory");
          else __syncthreads();
          c1_issue(c1, it + 2);
        }
      }
      if constexpr (s >= PF - 1) {
        constexpr int c = s - (PF - 1);
        constexpr int i = c / (3 * NKG), dx = (c / NKG) % 3, kg = c % NKG;
        // LDS operations complete in order: the wait for read c may leave outstanding everything issued after it -- the
        // younger reads AND the block-1 / window stores (asm, so their position is known) that went out after read c:
        // the 2 ring stores issued behind consume step C_STORE and the 3 * NXLD window stores behind C_XSTORE.
        constexpr int young_r = (NR - 1 - c) < (PF - 1) ? (NR - 1 - c) : (PF - 1);
        constexpr int young = young_r + ((c > C_STORE && c <= C_STORE + PF - 1 && c < NR) ? 2 : 0) +
                              ((c > C_XSTORE && c <= C_XSTORE + PF - 1 && c < NR) ? 3 * NXLD : 0) +
                              ((c > S_BAR - PF && c <= S_BAR) ? 2 : 0);      // the two window reads issued behind read S_BAR
        if constexpr (PIPE) lds_wait<young>(xbuf[c % PF]);
        const uint4 xv = __builtin_bit_cast(uint4, xbuf[c % PF]);
        if constexpr (i <= 2) acc0 = Mma<bf16_t>::run(w[i * 3 + dx][kg], xv, acc0);
        if constexpr (i >= 1) acc1 = Mma<bf16_t>::run(w[(i - 1) * 3 + dx][kg], xv, acc1);
        if constexpr (c == C_MFMA) c1_mfma(c1);
        if constexpr (c == C_XLOAD) x_load(it + 3);
        if constexpr (c == C_PSTORE) flush_pending();
        if constexpr (c == C_RELU) c1_relu(c1, it + 2);
        if constexpr (c == C_STORE) c1_store(c1, (PH + 2) % 3);
        if constexpr (c == C_XSTORE) x_store((it + 3) & 1);
        if constexpr (c == S_RELU0) {
#pragma unroll
          for (int e = 0; e < 16; ++e) acc0[e] = relu1(acc0[e], rlim);
        }
      }
    };
    {
      const unsigned ba = lds0 + BIAS2_OFF + (nsl * 32 + 4 * h) * 4;
      u32x4_t b0 = lds_frag<0, PIPE>(ba), b1 = lds_frag<32, PIPE>(ba), b2 = lds_frag<64, PIPE>(ba), b3 = lds_frag<96, PIPE>(ba);
      static_for(std::make_integer_sequence<int, PF - 1>{}, step);
      if constexpr (PIPE) lds_wait4<PF - 1>(b0, b1, b2, b3);
      const u32x4_t bq[4] = {b0, b1, b2, b3};
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc0[4 * g + e] = acc1[4 * g + e] = __uint_as_float(bq[g][e]);
    }
    static_for(std::make_integer_sequence<int, NR>{}, [&](auto s_c) {
      step(std::integral_constant<int, decltype(s_c)::value + PF - 1>{});
    });
    stamp(1);
    // AvgPool2d((2,1)) over the row pair (the 1/2 is in the weights), 16-byte packed stores
    const int Ho = H >> 1, to = t0 >> 1;
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = acc0[i] + relu1(acc1[i], rlim);
    po = obase + (size_t)to * (W * 64);
    pok = (to < Ho) && col_ok;
#pragma unroll
    for (int g = 0; g < 2; ++g) {          // channel groups (2g, 2g+1) -> after the swap lanes own 8 consecutive channels
      pq[4 * g] = pack_bf16x2(v[8 * g], v[8 * g + 1]);
      pq[4 * g + 1] = pack_bf16x2(v[8 * g + 2], v[8 * g + 3]);
      pq[4 * g + 2] = pack_bf16x2(v[8 * g + 4], v[8 * g + 5]);
      pq[4 * g + 3] = pack_bf16x2(v[8 * g + 6], v[8 * g + 7]);
    }
  };

  // Every iteration produces ring block it+2 and the windows of block it+3, also past the end of the image (the row
  // checks turn those into zeros that nobody reads): no wave-divergent or data-dependent branch in the loop.
  auto iteration = [&](auto ph_c, int it) {
    stamp(0);
    if (mg == 0) unit(ph_c, std::integral_constant<int, 0>{}, it);
    else unit(ph_c, std::integral_constant<int, 1>{}, it);
    stamp(2);
  };
  stamp(5);
  for (int it = it0; it < niter_seg; it += 3) {
    iteration(std::integral_constant<int, 0>{}, it);
    if (it + 1 < niter_seg) iteration(std::integral_constant<int, 1>{}, it + 1);
    if (it + 2 < niter_seg) iteration(std::integral_constant<int, 2>{}, it + 2);
  }
  flush_pending();                 // the last unit's outputs
#ifdef DFA_STAMPS
  if (lane == 0 && blockIdx.x < 2048) {
    long long* dd = g_diag12 + ((size_t)blockIdx.x * 4 + wave) * 8;
    for (int k = 0; k < 6; ++k) dd[k] = seg[k];
    dd[4] = __builtin_amdgcn_s_memrealtime() - r_begin;     // 100 MHz ticks: with the shader-clock lifetime -> the clock held
    dd[6] = t_begin;
    dd[7] = __builtin_amdgcn_s_memtime();
  }
#endif
}

// A operands of the block-1 MFMAs.  Lane (ch = lane&31, hh = lane>>5), element j: k = 8*hh + j, feature row dyy = k/4,
// tap dx = k%4 (dx = 3 is the zero pad).  even conv row: weight (dyy, dx) for dyy <= 2; odd: weight (dyy-1, dx), dyy >= 1.
__global__ void pack_conv1_mfma_kernel(const float* __restrict__ w1, const float* __restrict__ b1,
                                       uint4* __restrict__ c1pack, float* __restrict__ c1bias) {
  const int i = threadIdx.x;   // 256 threads: (operand k = i / 64, lane = i % 64)
  if (i < 32) c1bias[i] = 0.5f * b1[i];   // 16-byte aligned rows of 4: read as float4 by the kernel
  const int op = i >> 6, lane = i & 63, ch = lane & 31, hh = lane >> 5;
  const bool odd = op >= 2, lo = op & 1;
  bf16_t v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 8 * hh + j, dyy = k >> 2, dx = k & 3;
    const int dy = odd ? dyy - 1 : dyy;
    float wv = 0.f;
    if (dx < 3 && dy >= 0 && dy <= 2) wv = 0.5f * w1[ch * 9 + dy * 3 + dx];
    const bf16_t hi = float_to_bf16(wv);
    v[j] = lo ? float_to_bf16(wv - bf16_to_float(hi)) : hi;
  }
  c1pack[i] = *reinterpret_cast<const uint4*>(v);
}

hipError_t launch_pack_conv1_mfma(const float* w1, const float* b1, uint4* c1pack, float* c1bias, hipStream_t s) {
  hipLaunchKernelGGL(pack_conv1_mfma_kernel, dim3(1), dim3(256), 0, s, w1, b1, c1pack, c1bias);
  return hipGetLastError();
}

template <typename TX, bool PIPE>
static hipError_t launch_conv12_t(const Conv12Args& a, int B, hipStream_t s) {
  auto kern = conv12_fused_kernel<TX, PIPE>;
  static bool attr_set = false;
  static int lds_bytes = c12::LDS_BYTES;
  if (!attr_set) {
    // diagnostic: DFA_C12_LDS_PAD=<bytes> pads the dynamic LDS request (e.g. 60000 -> one workgroup per CU, one wave per SIMD)
    if (const char* pad = getenv("DFA_C12_LDS_PAD")) lds_bytes += atoi(pad);
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int niter = (a.H1 + c12::BR - 1) / c12::BR;
  const int nseg = a.seg_iters ? (niter + a.seg_iters - 1) / a.seg_iters : 1;
  hipLaunchKernelGGL(kern, dim3(B * a.nstrips, nseg), dim3(256), lds_bytes, s, a);
  return hipGetLastError();
}

hipError_t launch_conv12_fused(const void* x, int x_dtype, int64_t sb, int64_t st, int64_t sf, const uint4* c1pack,
                               const float* c1bias, const uint4* wpack2, const float* bias2, void* a2, int B, int T,
                               int F, hipStream_t s, int pipe, int seg_iters) {
  Conv12Args a{};
  a.seg_iters = seg_iters;
  a.x = x; a.sxb = sb; a.sxt = st; a.sxf = sf;
  a.c1pack = c1pack; a.c1bias = c1bias; a.wpack = wpack2; a.bias = bias2; a.out = (bf16_t*)a2;
  a.B = B; a.T = T; a.F = F; a.H1 = T / 2; a.nstrips = (F + c12::SW - 1) / c12::SW;
  hipError_t le;
  if (x_dtype == DFA_DTYPE_BF16) le = pipe ? launch_conv12_t<bf16_t, true>(a, B, s) : launch_conv12_t<bf16_t, false>(a, B, s);
  else le = pipe ? launch_conv12_t<float, true>(a, B, s) : launch_conv12_t<float, false>(a, B, s);
  if (le != hipSuccess) return le;
#ifdef DFA_STAMPS
  {
    static int calls = 0;
    if (++calls == 4000) {        // seconds of back-to-back launches: the clock has settled
      static long long hbuf[2048 * 4 * 8];
      hipDeviceSynchronize();
      hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(g_diag12), sizeof(hbuf));
      const int nw = (B * a.nstrips < 2048 ? B * a.nstrips : 2048) * 4;
      double m[8] = {0};
      for (int i = 0; i < nw; ++i) { for (int k = 0; k < 6; ++k) m[k] += hbuf[i * 8 + k]; m[6] += hbuf[i * 8 + 7] - hbuf[i * 8 + 6]; }
      fprintf(stderr, "[stamps conv12] in-kernel clock %.3f GHz (shader cycles / 100 MHz real-time ticks, mean over waves)\n", m[6] / (m[4] * 10.0));
      fprintf(stderr, "[stamps conv12] waves %d  mean cycles/wave: x_load %.0f  mfma_stream %.0f  epilogue %.0f  barrier %.0f  prologue %.0f  lifetime %.0f\n",
              nw, m[0] / nw, m[1] / nw, m[2] / nw, m[3] / nw, m[5] / nw, m[6] / nw);
    }
  }
#endif
  return hipGetLastError();
}

}  // namespace dfa
