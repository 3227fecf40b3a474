// train_conv1_mfma.hip -- the three train-mode passes over the 1-channel first block of the CNN2D (src/model.py:15-19, as run
// by src/train.py:71-76) on the matrix cores, bf16 features.  train_conv1.hip / conv1.hip do the same passes on the vector
// ALU (9 + ~15 FMAs per pixel and channel: 0.30 + 0.31 + 0.74 ms at [256,321,180]); they stay for fp32, for fp32 feature
// tensors and for batches with the augmentation folded in, and as the twins the GPU tests compare this file with.
//
// One im2col tile serves every matrix product.  A workgroup walks NT = 2 pooled rows (4 convolution rows) x all F columns per
// step: the feature rows go to LDS once, then every pixel gets a 32-byte record  col[pixel] = {x taps 0..8, 1.0, 0 x 6}  (bf16, two records per 80-byte slot;
// all zeros outside the image, so such pixels drop out of every sum by themselves).
//   product 1   y[32 ch x 32 px] = W[32 x 16] . col^T[16 x 32]     v_mfma_f32_32x32x16_bf16, W = hi + lo + lo2: three bf16 terms
//               carry the fp32 weight exactly (three MFMAs, x is bf16: every product is exact, only the fp32 accumulation
//               rounds -- with two terms 0.1 % of the bf16 a1 moved by an ulp against the fp32 kernels and the tiny-batch
//               oracle test drifted); the bias rides on the 1.0 slot.  The X
//               operand is one aligned ds_read_b128 of a pixel's record.  Result: lane = pixel, 16 channels in registers.
//               FWD takes it as y[ch][px] (lane = pixel, 16 channels in registers: the layout a1 is stored in); STATS and BWD swap
//               the operands, y^T[px][ch] (lane = channel, 16 pixels in registers) -- per-channel sums then need no cross-lane
//               work, and the masked upstream gradient is already the register image of product 2's B operand.
//   product 2   G[taps x 32 ch] += col^T[taps x 16 px] . dy[16 px x 32 ch]   v_mfma_f32_32x32x16_bf16; the col^T operand comes from
//               two ds_read_b64_tr_b16 (pixels-major records read as 4 + 4 pixels per lane for one tap), in the pixel order the
//               accumulator registers of product 1 hold: k = 8h + e  <->  pixel 16j + 4h + (e & 3) + 8 (e >> 2).  Only 16 of the
//               32 tap rows exist; rows 16..31 of G are never read.
// Modes:
//   STATS  raw weights: z = conv1(x) + b;  per-channel sum / sum of squares over all T rows (BatchNorm sees the odd last row),
//          and XX = col^T col (v_mfma_f32_16x16x32_bf16, both operands the same transposed read): the 9 x 9 tap moments and the
//          tap sums the fused backward algebra needs (train_conv1.hip header) -- same records as conv1_train_kernel<STATS_XX>.
//   FWD    BatchNorm-folded weights: y -> ReLU -> AvgPool2d((2,1)) -> Dropout -> a1 [B][Ho][F][32] bf16 (two 16-byte stores per
//          lane after a half-wave exchange, as conv12_fused.hip).
//   BWD    the same y (same operand registers: the forward's ReLU mask); dy = mask * da1 is exact in bf16 -- the dropout keep
//          mask is applied where da1 is produced (conv_split.hip, ConvArgs::drop) and the factor 0.5 * dropout scale goes on the
//          sums -- so A[c][k] = sum dy * x_k and S1 = sum dy (the 1.0 tap) are two product 2's per 32 pixels and row.  da1 is
//          read as 2-byte elements (lane = channel: 64 contiguous bytes per 32 lanes).  S2 = sum dy * xhat needs no pass of its own: xhat = is*(z - mu) and
//          z = b + sum_k w_k x_k, so  sum dy*z = b*S1 + sum_k w_k A[c][k]  (conv1_bwd_finalize_kernel, derive_s2).
#include "dfa_internal.h"
#include "rng.h"
#include <stdlib.h>

namespace dfa {

namespace {
constexpr int NT = 2, NR = 2 * NT + 2, NCR = 2 * NT;   // pooled rows per step; feature rows / convolution rows in LDS
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32x2_t lds_tr16(unsigned addr) {
  return __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(size_t)addr));
}
__device__ __forceinline__ f32x16_t mma32(const uint4& w, const uint4& x, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w), __builtin_bit_cast(bf16x8_t, x), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mma16(const uint4& a, const uint4& b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
}  // namespace

// bytes per feature row in LDS: (FP + 2) bf16, padded to 3 (mod 32) dwords -- the 32 lanes of a ds_write_b16 group hold (row
// 0..5, column pair) combinations, which then fall on distinct banks
__host__ __device__ inline int c1x_row_bytes(int FP) {
  const int d = (FP + 2) / 2;
  return 4 * (d + (((3 - d) % 32) + 32) % 32);
}

struct C1xArgs {
  const bf16_t* x;
  int64_t sb, st, sf;
  const float* w;       // [32][9]  raw (STATS) or BatchNorm-folded (FWD, BWD)
  const float* b;       // [32]
  bf16_t* a1;           // FWD: out [B][Ho][F][32]
  const bf16_t* da1;    // BWD: in  [B][Ho][F][32]
  float* partial;       // STATS: [nblk][32][2] then [nblk][96];  BWD: [nblk][32][11]
  int T, F, Ho, FP, rows_per_wg;
  int poolw;            // BWD: da1 is [B][Ho][F / poolw][32]: 1 = AvgPool2d((2,1)) (CNN2D), 2 = AvgPool2d(2) (auto-encoder block 1, F even)
  float out_scale;      // 0.5 * dropout scale: FWD folds it into the weights, BWD applies it to the sums
  DropCfg dc;
  int dbg;              // DFA_C1X_DBG (diagnostic): 1 = no tiles, 2 = no im2col, 4 = no feature-row traffic
};

template <int MODE>
__global__ __launch_bounds__(256, 2) void conv1_mfma_kernel(C1xArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int i16 = lane & 15, q4 = lane >> 4, qrow = i16 >> 2, pq = i16 & 3;
  const int T = a.T, F = a.F, FP = a.FP, NFC = FP >> 5, Ho = a.Ho;
  const int RS = c1x_row_bytes(FP);                  // feature row in LDS: element i <-> f = i - 1
  char* raw = smem;                                  // [NR][RS]
  // im2col records: a PAIR of adjacent pixels takes an 80-byte slot (2 x 32 bytes + 16 of padding): the pair's thread writes
  // its four 16-byte pieces with ds_write_b128, whose 8-lane groups then fall on disjoint banks (at the dense 64-byte pitch they
  // were 4-way conflicted: 75 % of the LDS cycles of these kernels were conflict cycles, SQ_LDS_BANK_CONFLICT)
  const int ROWB = FP * 40;                          // bytes per convolution row of records
  char* col = smem + ((NR * RS + 15) & ~15);         // [NCR][FP / 2] x 80 bytes
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned col0 = lds0 + (unsigned)(col - smem);
  const int b = blockIdx.y;
  const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  const int NPm = (MODE == C1X_STATS) ? (T + 1) / 2 : Ho;
  const int to_begin = blockIdx.x * a.rows_per_wg, to_end = min(NPm, to_begin + a.rows_per_wg);
  const bf16_t* xb = a.x + (int64_t)b * a.sb;
  const bool t_fast = (a.st == 1);

  // A operand of product 1: lane (channel r, half h), element e <-> k = 8h + e: taps 0..8, bias at k = 9, zeros behind
  uint4 whi, wlo, wl2;
  {
    unsigned hi[4], lo[4], l2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int k = 8 * h + 2 * j + u;
        v[u] = (k < 9) ? a.w[r * 9 + k] : (k == 9 ? a.b[r] : 0.f);
        if constexpr (MODE == C1X_FWD) v[u] *= a.out_scale;     // relu(s*y) = s*relu(y): the pool's 1/2 and the dropout scale ride on the weights
      }
      const float h0 = bf16_to_float(float_to_bf16(v[0])), h1 = bf16_to_float(float_to_bf16(v[1]));
      const float m0 = bf16_to_float(float_to_bf16(v[0] - h0)), m1 = bf16_to_float(float_to_bf16(v[1] - h1));
      hi[j] = pack_bf16x2(v[0], v[1]);
      lo[j] = pack_bf16x2(v[0] - h0, v[1] - h1);
      l2[j] = pack_bf16x2((v[0] - h0) - m0, (v[1] - h1) - m1);      // three bf16 terms = the 24-bit mantissa: the fp32 weight, exactly
    }
    whi = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    wlo = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    wl2 = make_uint4(l2[0], l2[1], l2[2], l2[3]);
  }

  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  f32x2_t sa = {0.f, 0.f}, sq = {0.f, 0.f};   // STATS: this lane's channel r: sum / sum of squares (two interleaved partial sums)
  f32x4_t gxx = {0.f, 0.f, 0.f, 0.f};         // STATS: XX
  f32x16_t gw;                                // BWD: G[tap][channel r]
#pragma unroll
  for (int i = 0; i < 16; ++i) gw[i] = 0.f;

  // feature rows of a step -> registers (issued before the previous step's tiles are computed), then -> LDS.  Which element a
  // thread moves does not depend on the step: its global offset, LDS offset and column validity are computed once.
  constexpr int NXR = 6;                      // NR * (FP + 2) <= 6 * 256 elements: FP <= 254 (checked by the launcher)
  unsigned short xr[NXR];
  constexpr int64_t XG_NONE = INT64_MIN;      // (a real offset can be -1: row -1 of column 0)
  int64_t xg[NXR];                            // element offset of (row rr, column f) relative to row 2*to0, or XG_NONE: never loaded
  int xl[NXR], xrr[NXR];                      // LDS byte offset (-1: not this thread's), row rr
#pragma unroll
  for (int k = 0; k < NXR; ++k) {
    const int e = k * 256 + tid;
    int rr, cc;
    if (t_fast) { cc = e / NR; rr = e - cc * NR; } else { rr = e / (FP + 2); cc = e - rr * (FP + 2); }
    const int f = cc - 1;
    const bool mine = e < NR * (FP + 2);
    xrr[k] = rr;
    xl[k] = mine ? rr * RS + cc * 2 : -1;
    xg[k] = (mine && f >= 0 && f < F) ? (int64_t)(rr - 1) * a.st + (int64_t)f * a.sf : XG_NONE;
  }
  auto raw_load = [&](int to0) {
    const bf16_t* xs = xb + (int64_t)(2 * to0) * a.st;
#pragma unroll
    for (int k = 0; k < NXR; ++k) {
      const int t = 2 * to0 - 1 + xrr[k];
      unsigned short v = 0;
      if (xg[k] != XG_NONE && t >= 0 && t < T) v = xs[xg[k]].v;
      xr[k] = v;
    }
  };
  auto raw_store = [&]() {
#pragma unroll
    for (int k = 0; k < NXR; ++k)
      if (xl[k] >= 0) *(unsigned short*)(raw + xl[k]) = xr[k];
  };
  // im2col: a thread's (up to two) pixel pairs, fixed over the steps
  constexpr int NIC = 2;                      // NCR * FP / 2 <= 2 * 256 pairs: FP <= 256
  int ic_src[NIC], ic_dst[NIC], ic_tr[NIC], ic_fi[NIC];
#pragma unroll
  for (int k = 0; k < NIC; ++k) {
    const int e = k * 256 + tid;
    const int tr = e / (FP >> 1), fi = 2 * (e - tr * (FP >> 1));
    const bool mine = e < NCR * (FP >> 1);
    ic_tr[k] = mine ? tr : -1;
    ic_fi[k] = fi;
    ic_src[k] = tr * RS + fi * 2;
    ic_dst[k] = tr * ROWB + (fi >> 1) * 80;
  }

  if (to_begin < to_end) raw_load(to_begin);
  for (int to0 = to_begin; to0 < to_end; to0 += NT) {
    if (!(a.dbg & 4)) raw_store();                 // the previous step's im2col (the only reader of the feature rows) is behind its barrier
    __syncthreads();             // feature rows complete; every wave is done with the previous step's records
    // ---- im2col records, two adjacent pixels per thread: pixel (convolution row tr, column fi) <- rows tr..tr+2, elements
    //      fi..fi+2 of the feature rows (element i <-> f = i - 1)
#pragma unroll
    for (int k = 0; k < NIC; ++k) {
      const int tr = ic_tr[k], fi = ic_fi[k];
      if (tr < 0 || (a.dbg & 2)) continue;
      const bool row_ok = 2 * to0 + tr < T && to0 + (tr >> 1) < to_end;
      unsigned d0[3], d1[3];        // row dy: d0 = elements (fi, fi+1), d1 = (fi+2, fi+3); pixel fi uses fi..fi+2, pixel fi+1 uses fi+1..fi+3
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const unsigned* rp = (const unsigned*)(raw + ic_src[k] + dy * RS);
        d0[dy] = rp[0];
        d1[dy] = rp[1];
      }
      const unsigned m0[3] = {__builtin_amdgcn_alignbit(d1[0], d0[0], 16), __builtin_amdgcn_alignbit(d1[1], d0[1], 16),
                              __builtin_amdgcn_alignbit(d1[2], d0[2], 16)};             // elements (fi+1, fi+2) of each row
      const bool ok0 = row_ok && fi < F, ok1 = row_ok && fi + 1 < F;
      // record = taps (row 0: 0 1 2)(row 1: 3 4 5)(row 2: 6 7 8), then 1.0, then zeros
      uint4 a0, a1;
      a0.x = d0[0];
      a0.y = (d1[0] & 0xffffu) | (d0[1] << 16);
      a0.z = m0[1];
      a0.w = d0[2];
      a1.x = m0[0];
      a1.y = (d1[0] >> 16) | (m0[1] << 16);
      a1.z = d1[1];
      a1.w = m0[2];
      unsigned t0 = (d1[2] & 0xffffu) | 0x3f800000u, t1 = (d1[2] >> 16) | 0x3f800000u;      // tap 8, then 1.0
      if (!ok0) { a0 = make_uint4(0u, 0u, 0u, 0u); t0 = 0u; }
      if (!ok1) { a1 = make_uint4(0u, 0u, 0u, 0u); t1 = 0u; }
      char* dst = col + ic_dst[k];
      *(uint4*)dst = a0;
      *(uint4*)(dst + 16) = make_uint4(t0, 0u, 0u, 0u);
      *(uint4*)(dst + 32) = a1;
      *(uint4*)(dst + 48) = make_uint4(t1, 0u, 0u, 0u);
    }
    __syncthreads();
    if (to0 + NT < to_end && !(a.dbg & 4)) raw_load(to0 + NT);      // in flight under the tiles
    // ---- tiles: (pooled row, 32-column chunk), round-robin over the four waves
    for (int id = wave; id < NT * NFC; id += 4) {
      const int tl = id / NFC, fc = id - tl * NFC;
      const int to = to0 + tl;
      if (to >= to_end || (a.dbg & 1)) continue;
      const int f0 = fc * 32;
      const unsigned cbe = (unsigned)((2 * tl) * ROWB + f0 * 40), cbo = cbe + (unsigned)ROWB;
      const unsigned xoff = (unsigned)((r >> 1) * 80 + (r & 1) * 32 + 16 * h);
      const uint4 xe = *(const uint4*)(col + cbe + xoff), xo = *(const uint4*)(col + cbo + xoff);
      f32x16_t ye, yo;
#pragma unroll
      for (int i = 0; i < 16; ++i) { ye[i] = 0.f; yo[i] = 0.f; }
      if constexpr (MODE == C1X_FWD) {       // y[ch][px]: lane = pixel r, registers = channels (i&3) + 8*(i>>2) + 4h
        ye = mma32(wl2, xe, ye);             // smallest terms first
        yo = mma32(wl2, xo, yo);
        ye = mma32(wlo, xe, ye);
        yo = mma32(wlo, xo, yo);
        ye = mma32(whi, xe, ye);
        yo = mma32(whi, xo, yo);
      } else {                               // y^T[px][ch]: lane = channel r, registers = pixels (i&3) + 8*(i>>2) + 4h
        ye = mma32(xe, wl2, ye);
        yo = mma32(xo, wl2, yo);
        ye = mma32(xe, wlo, ye);
        yo = mma32(xo, wlo, yo);
        ye = mma32(xe, whi, ye);
        yo = mma32(xo, whi, yo);
      }

      if constexpr (MODE == C1X_STATS) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const f32x2_t ve = {ye[i], ye[i + 1]}, vo = {yo[i], yo[i + 1]};
          sa += ve;
          sq = __builtin_elementwise_fma(ve, ve, sq);
          sa += vo;
          sq = __builtin_elementwise_fma(vo, vo, sq);
        }
        const int px = 8 * q4 + qrow;                                          // 16 x 16 x 32: lane (tap i16, q4) <- pixels 8*q4 .. +7
        const unsigned tr_off = (unsigned)((px >> 1) * 80 + (px & 1) * 32 + pq * 8);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const unsigned ad = col0 + (p ? cbo : cbe) + tr_off;
          const u32x2_t t0 = lds_tr16(ad), t1 = lds_tr16(ad + 160);          // + 4 pixels = 2 pair slots
          const uint4 op = make_uint4(t0[0], t0[1], t1[0], t1[1]);
          gxx = mma16(op, op, gxx);
        }
      } else if constexpr (MODE == C1X_FWD) {
        unsigned pk[8];               // pk[2G + jj]: channels 8G + 4h + 2jj, +1 (already scaled: see the weight operands)
#pragma unroll
        for (int i = 0; i < 16; i += 2)
          pk[i >> 1] = pack_bf16x2(fmaxf(ye[i], 0.f) + fmaxf(yo[i], 0.f), fmaxf(ye[i + 1], 0.f) + fmaxf(yo[i + 1], 0.f));
        const int f = f0 + r;
        const size_t pix = ((size_t)b * Ho + to) * F + f;
#pragma unroll
        for (int g = 0; g < 4; g += 2) {       // half-wave exchange: this lane ends with the 8 channels of octet g + h
          const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * g], pk[2 * g + 2], false, false);
          const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * g + 1], pk[2 * g + 3], false, false);
          uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          const int oct = g + h;
          if (a.dc.thresh != 0) {              // dropped elements -> 0 (the survivors' scale is in the weights)
            unsigned km[4];
            drop_keep8(a.dc, pix * 32 + oct * 8, km);
            o.x &= km[0]; o.y &= km[1]; o.z &= km[2]; o.w &= km[3];
          }
          if (f < F) *(uint4*)(a.a1 + pix * 32 + oct * 8) = o;
        }
      } else {
        // upstream gradient of channel r at this lane's 16 pixels.  Columns beyond F are clamped to a valid address: their im2col
        // records are all zero (y = 0 -> mask off, taps = 0), whatever is read there drops out.
        const int psh = a.poolw - 1;                                         // pooled column = pixel column >> psh
        const bf16_t* drow = a.da1 + ((size_t)b * Ho + to) * (F >> psh) * 32 + r;
        unsigned dpk[8];
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          const int p0 = min(f0 + (i & 3) + 8 * (i >> 2) + 4 * h, F - 1), p1 = min(f0 + (i & 3) + 8 * (i >> 2) + 4 * h + 1, F - 1);
          dpk[i >> 1] = (unsigned)drow[(size_t)(p0 >> psh) * 32].v | ((unsigned)drow[(size_t)(p1 >> psh) * 32].v << 16);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int px = 4 * h + qrow;
          const unsigned cb = col0 + (p ? cbo : cbe) + (unsigned)((px >> 1) * 80 + (px & 1) * 32 + pq * 8);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            unsigned dv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const float y0 = p ? yo[8 * j + 2 * u] : ye[8 * j + 2 * u], y1 = p ? yo[8 * j + 2 * u + 1] : ye[8 * j + 2 * u + 1];
              dv[u] = (y0 > 0.f ? (dpk[4 * j + u] & 0x0000ffffu) : 0u) | (y1 > 0.f ? (dpk[4 * j + u] & 0xffff0000u) : 0u);
            }
            // col^T for pixels 16j + 4h + {0..3} and 16j + 8 + 4h + {0..3}: the K order of the registers above
            const u32x2_t t0 = lds_tr16(cb + (unsigned)(16 * j * 40)), t1 = lds_tr16(cb + (unsigned)((16 * j + 8) * 40));
            gw = mma32(make_uint4(t0[0], t0[1], t1[0], t1[1]), make_uint4(dv[0], dv[1], dv[2], dv[3]), gw);
          }
        }
      }
    }
  }

  // ---- block records
  __syncthreads();
  float* red = (float*)smem;
  if constexpr (MODE == C1X_STATS) {
    float s1v = sa[0] + sa[1], s2v = sq[0] + sq[1];
    s1v += __shfl_xor(s1v, 32, 64);
    s2v += __shfl_xor(s2v, 32, 64);
    if (h == 0) { red[wave * 64 + r * 2] = s1v; red[wave * 64 + r * 2 + 1] = s2v; }
    float* red2 = red + 256;                   // [4 waves][16 j][16 k]
#pragma unroll
    for (int e = 0; e < 4; ++e) red2[wave * 256 + (4 * q4 + e) * 16 + i16] = gxx[e];
    __syncthreads();
    if (tid < 64) a.partial[blk * 64 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
    if (tid < 90) {
      const int j = tid < 81 ? tid / 9 : 9, k = tid < 81 ? tid - 9 * (tid / 9) : tid - 81;
      const int o = j * 16 + k;
      float* part2 = a.partial + (size_t)gridDim.x * gridDim.y * 64;
      part2[blk * 96 + tid] = (red2[o] + red2[256 + o]) + (red2[512 + o] + red2[768 + o]);
    }
  } else if constexpr (MODE == C1X_BWD) {
    // gw[i]: tap (i&3) + 8*(i>>2) + 4h (i < 8: taps 0..15), channel r  ->  red[wave][channel][12 taps]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int tap = (i & 3) + 8 * (i >> 2) + 4 * h;
      if (tap < 12) red[wave * 384 + r * 12 + tap] = gw[i];
    }
    __syncthreads();
    for (int e = tid; e < 352; e += 256) {
      const int ch = e / 11, k = e - 11 * ch;
      float v = 0.f;
      if (k < 10) {
        const int o = ch * 12 + k;
        v = ((red[o] + red[384 + o]) + (red[768 + o] + red[1152 + o])) * a.out_scale;
      }
      a.partial[blk * 352 + e] = v;          // k = 10 (S2) is derived by conv1_bwd_finalize_kernel
    }
  }
}

int conv1_mfma_rows_per_wg(int B, int T, int F) {
  // 12 workgroups per utterance at F = 180 (3072 at B = 256: three rounds of 4 resident workgroups per CU).  One round of 4 long
  // workgroups per utterance with the backward forced to 128 VGPRs measured slower (0.37 vs 0.22 ms: it spills).
  const int nb_max = conv1_train_blocks(B, T, F) / B;      // the partial buffer is sized for that many records per utterance
  const int np = (T + 1) / 2;
  int rows = (np + nb_max - 1) / nb_max;
  rows = (rows + NT - 1) / NT * NT;
  return rows;
}
int conv1_mfma_blocks(int B, int T, int F) {
  const int rows = conv1_mfma_rows_per_wg(B, T, F), np = (T + 1) / 2;
  return B * ((np + rows - 1) / rows);
}

hipError_t launch_conv1_mfma(int mode, const void* x, int64_t sb, int64_t st, int64_t sf, const float* w, const float* bias,
                             void* a1, const void* da1, float* partial, int B, int T, int F, const DropCfg& dc, hipStream_t s, int poolw) {
  C1xArgs a{};
  if (poolw != 1 && (poolw != 2 || mode != C1X_BWD || (F & 1))) return hipErrorInvalidValue;   // the 2x2 form exists for the backward pass (STATS is pool-free)
  a.poolw = poolw;
  a.x = (const bf16_t*)x; a.sb = sb; a.st = st; a.sf = sf;
  a.w = w; a.b = bias; a.a1 = (bf16_t*)a1; a.da1 = (const bf16_t*)da1; a.partial = partial;
  a.T = T; a.F = F; a.Ho = T / 2; a.FP = (F + 31) / 32 * 32;
  a.rows_per_wg = conv1_mfma_rows_per_wg(B, T, F);
  a.dc = dc;
  { static const char* e = getenv("DFA_C1X_DBG"); a.dbg = e ? atoi(e) : 0; }
  a.out_scale = (poolw == 2 ? 0.25f : 0.5f) * (dc.thresh != 0 ? dc.scale : 1.0f);
  const int np = (T + 1) / 2;
  dim3 grid((np + a.rows_per_wg - 1) / a.rows_per_wg, B), block(256);
  const size_t RS = (size_t)c1x_row_bytes(a.FP);
  size_t lds = ((NR * RS + 15) & ~(size_t)15) + (size_t)NCR * a.FP * 40;
  if (lds < 8192) lds = 8192;                  // the block-record reduction reuses the front of the buffer
  if (lds > 64 * 1024 || NR * (a.FP + 2) > 6 * 256 || NCR * (a.FP / 2) > 2 * 256) return hipErrorInvalidValue;
  if (mode == C1X_STATS) hipLaunchKernelGGL(conv1_mfma_kernel<C1X_STATS>, grid, block, lds, s, a);
  else if (mode == C1X_FWD) hipLaunchKernelGGL(conv1_mfma_kernel<C1X_FWD>, grid, block, lds, s, a);
  else hipLaunchKernelGGL(conv1_mfma_kernel<C1X_BWD>, grid, block, lds, s, a);
  return hipGetLastError();
}

}  // namespace dfa
