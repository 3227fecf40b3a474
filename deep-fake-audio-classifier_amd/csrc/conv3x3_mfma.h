// conv3x3_mfma.h -- 3x3 / pad 1 convolution as an implicit GEMM on the gfx950 matrix cores, with the
// BatchNorm(eval, folded) + ReLU + pooling / time-mean epilogue fused in.
//
// Replaces, per launch, one "Conv2d -> BatchNorm2d -> ReLU [-> AvgPool2d]" group of the reference:
//   src/model.py:21-25 (block 2), :27-29 + :37 (block 3 + mean over T),
//   src/model_cae.py:40-55 (encoder blocks 2-4).
//
// Data layout (HBM): activations are channels-last  act[b][t][f][c]  so that one tap of one pixel is
// CIN contiguous elements; weights are pre-packed by pack.hip in the exact register order the MFMA
// B-operand wants:  wpack[cout/32][tap][kgroup][lane] (16 bytes each).
//
// Work decomposition
//   workgroup  = (utterance b, strip of 32*MT feature columns, chunk of 32*NSL output channels); it walks
//                DOWN the time axis keeping a ring of input rows in LDS, so every input element is read from
//                HBM once per strip (+2 halo columns) and there is no vertical halo re-read.
//   wave       = (N-slice nsl of 32 output channels, M-group mg).  The wave keeps its full 9 x CIN x 32 weight
//                slice in VGPRs for the whole kernel (B operand), and streams A fragments from the LDS ring.
//   iteration  = 2*RP output rows (RP row pairs).  A "unit" is one pair of 32-pixel M tiles (rows t, t+1; same
//                32 columns): both accumulators share the A fragments of the two input rows they have in
//                common, so a unit issues 12*NKG ds_read_b128 for 18*NKG k-groups of MFMA work.
//   staging    = block j of the ring holds input rows [2RP*j-1, 2RP*(j+1)-1).  Iteration `it` computes from
//                blocks it, it+1 while block it+2 is fetched (global -> VGPR before the MFMAs, VGPR -> LDS after
//                them), one __syncthreads() per iteration.
//
// LDS image: pixel p (linear index ring_row*SLOTS + slot) owns PB = CIN*sizeof(T) bytes = CPP 16-byte chunks;
// chunk c is stored at physical chunk  c ^ swz(p)  so that the 16 lanes of every ds_read_b128 lane group
// (consecutive pixels, same logical chunk) hit 16 different 16-byte bank slots (MI355X_MICROARCH "LDS").
//
// MFMA: bf16 -> v_mfma_f32_32x32x16_bf16 (A: 8 bf16 per lane = one 16-byte chunk);
//       f32  -> v_mfma_f32_32x32x2_f32  (exact fp32 fma chain; one 16-byte chunk feeds 4 MFMAs).
//       Lane (r = lane&31, h = lane>>5) reads logical chunk 2*kg+h of pixel r (+tap shift), i.e. input channels
//       KG*kg + (KG/2)*h + j.  C/D: column (= output channel) lane&31, row (= pixel) (i&3)+8*(i>>2)+4*h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
template <typename T>
__device__ __forceinline__ T cvt_out(float f);
template <>
__device__ __forceinline__ float cvt_out<float>(float f) { return f; }
template <>
__device__ __forceinline__ bf16_t cvt_out<bf16_t>(float f) { return float_to_bf16(f); }

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
  const float* acc_in; // ACCIN: [B][H][W][COUT] fp32 partial sums to start from
  float* raw_out;      // EPI_RAW: [B][H][W][COUT] fp32 partial sums (no bias, no activation)
  // train-mode BatchNorm statistics (EPI_PLAIN only): per-workgroup partial sums of the stored values v and v*v
  // per output channel, partial[(blockIdx.x * COUT + channel) * 2 + {0,1}]; reduced in a fixed order by bn_finalize.
  float* stats_partial;
  const void* zero_page;  // DMA staging: >= 16 zero bytes in device memory (source of out-of-image chunks)
};

template <int PB>
__device__ __forceinline__ int lds_swz(int p) {
  if (PB == 64) return (p >> 2) & 3;
  if (PB == 128) return (p >> 1) & 7;
  return p & 15;  // 256, 512
}

template <typename T>
struct Mma;
template <>
struct Mma<bf16_t> {
  static __device__ __forceinline__ f32x16_t run(const uint4& a, const uint4& b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c,
                                                   0, 0, 0);
  }
};
template <>
struct Mma<float> {
  static __device__ __forceinline__ f32x16_t run(const uint4& a, const uint4& b, f32x16_t c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
  }
};

template <typename T, int CIN, int NSL, int MG, int RP, int MT, int EPI>
struct ConvCfg {
  static constexpr int ES = sizeof(T);
  static constexpr int KG = 32 / ES;       // input channels per k-group (two 16-byte chunks)
  static constexpr int NKG = CIN / KG;
  static constexpr int PB = CIN * ES;      // bytes per pixel
  static constexpr int CPP = PB / 16;      // 16-byte chunks per pixel
  static constexpr int SLOTS = 32 * MT + 2;
  static constexpr int BR = 2 * RP;        // rows per ring block
  static constexpr int NT = 64 * NSL * MG;
  static constexpr int NCH = BR * SLOTS * CPP;
  static constexpr int NLD = (NCH + NT - 1) / NT;
  static constexpr int UNITS = RP * MT;
  static constexpr int UPW = UNITS / MG;   // units per wave per iteration
  static constexpr int RING_BYTES = 3 * BR * SLOTS * PB;
  static constexpr int EPI_LDW = 32 * MT + 1;
  static constexpr int EPI_BYTES = (EPI == EPI_MEAN_T) ? NSL * 32 * EPI_LDW * 4 : 0;
  static constexpr int LDS_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
  static_assert(UNITS % MG == 0, "units must split evenly over the M groups");
  static_assert((BR & (BR - 1)) == 0, "rows per block must be a power of two");
  static_assert(EPI != EPI_MEAN_T || MG == 1, "time-mean epilogue keeps column sums per wave: MG must be 1");
  static_assert(CIN % KG == 0, "CIN must be a multiple of the k-group");
};

template <typename T, int CIN, int NSL, int MG, int RP, int MT, int EPI, int MINW, bool ACCIN = false, bool DMA = false>
__global__ __launch_bounds__(64 * NSL * MG, MINW) void conv3x3_mfma_kernel(ConvArgs a) {
  using C = ConvCfg<T, CIN, NSL, MG, RP, MT, EPI>;
  constexpr int PB = C::PB, CPP = C::CPP, SLOTS = C::SLOTS, BR = C::BR, NT = C::NT, NKG = C::NKG;
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
  const int f0 = strip * (32 * MT);
  const int H = a.H, W = a.W, COUT = a.COUT;
  const int cout_base = blockIdx.y * (NSL * 32);
  const int n = cout_base + nsl * 32 + r;  // this lane's output channel

  const int ipb = a.in_pix_bytes ? a.in_pix_bytes : PB;
  const char* in_b = (const char*)a.in + (size_t)b * H * W * ipb + a.in_ch_off_bytes;

  // ---- weights: the wave's [9][NKG] 16-byte B fragments stay in registers for the whole kernel
  uint4 w[9][NKG];
  {
    const uint4* wp = a.wpack + ((size_t)(blockIdx.y * NSL + nsl) * 9 * NKG) * 64 + lane;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kg = 0; kg < NKG; ++kg) w[tap][kg] = wp[(tap * NKG + kg) * 64];
  }
  const float bv = a.bias[n];

  // ---- staging of ring block j: input rows t = BR*j - 1 + rowi, slots f = f0 - 1 + slot
  uint4 stg[C::NLD];
  auto stage_load = [&](int j) {
#pragma unroll
    for (int k = 0; k < C::NLD; ++k) {
      const int g = k * NT + tid;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (g < C::NCH) {
        const int rowi = g / (SLOTS * CPP);
        const int rem = g - rowi * (SLOTS * CPP);
        const int slot = rem / CPP, c = rem % CPP;
        const int t = BR * j - 1 + rowi, f = f0 - 1 + slot;
        if (t >= 0 && t < H && f >= 0 && f < W) v = *(const uint4*)(in_b + ((size_t)t * W + f) * ipb + c * 16);
      }
      stg[k] = v;
    }
  };
  auto stage_store = [&](int ringblk) {
#pragma unroll
    for (int k = 0; k < C::NLD; ++k) {
      const int g = k * NT + tid;
      if (g < C::NCH) {
        const int rowi = g / (SLOTS * CPP);
        const int rem = g - rowi * (SLOTS * CPP);
        const int slot = rem / CPP, c = rem % CPP;
        const int p = (ringblk * BR + rowi) * SLOTS + slot;
        *(uint4*)(smem + p * PB + ((c ^ lds_swz<PB>(p)) << 4)) = stg[k];
      }
    }
  };

  // LDS-DMA variant (global_load_lds_dwordx4): no staging VGPRs, no ds_write.  One wave instruction fills 64
  // consecutive PHYSICAL 16-byte chunks (1 KiB, wave-uniform LDS base + lane*16), so the chunk swizzle is applied to
  // the per-lane SOURCE address: the lane that owns physical chunk c' of pixel p fetches logical chunk c' ^ swz(p).
  // Out-of-image chunks read a 16-byte zero page.  The transfers are retired by the vmcnt(0) hipcc emits at the
  // iteration's __syncthreads().
  auto stage_dma = [&](int j, int ringblk) {
#pragma unroll
    for (int k = 0; k < C::NLD; ++k) {
      const int g = k * NT + tid;
      if (g < C::NCH) {
        const int pix = g / CPP, cph = g % CPP;
        const int rowi = pix / SLOTS, slot = pix - rowi * SLOTS;
        const int p = ringblk * BR * SLOTS + pix;
        const int c = cph ^ lds_swz<PB>(p);
        const int t = BR * j - 1 + rowi, f = f0 - 1 + slot;
        const char* src = (t >= 0 && t < H && f >= 0 && f < W) ? in_b + ((size_t)t * W + f) * ipb + c * 16
                                                               : (const char*)a.zero_page;
        char* dst = smem + ((size_t)ringblk * BR * SLOTS * CPP + k * NT + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };

  float cs[EPI == EPI_MEAN_T ? MT : 1][16];
  if (EPI == EPI_MEAN_T) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) cs[m][i] = 0.f;
  }

  float st1 = 0.f, st2 = 0.f;  // EPI_PLAIN: running sum / sum of squares of this lane's channel
  const int niter = (H + BR - 1) / BR;
  if (DMA) {
    stage_dma(0, 0);
    stage_dma(1, 1);
  } else {
    stage_load(0);
    stage_store(0);
    stage_load(1);
    stage_store(1);
  }
  __syncthreads();

  int blk0 = 0;  // it % 3
  for (int it = 0; it < niter; ++it) {
    const bool pf = (it + 1 < niter);
    const int blk1 = (blk0 == 2) ? 0 : blk0 + 1;
    const int blk2 = (blk1 == 2) ? 0 : blk1 + 1;
    if (pf) {
      if (DMA) stage_dma(it + 2, blk2); else stage_load(it + 2);
    }

#pragma unroll
    for (int uu = 0; uu < C::UPW; ++uu) {
      const int u = uu * MG + mg;
      const int rp = u / MT, m = u - rp * MT;
      f32x16_t acc0, acc1;
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
      if (ACCIN) {
        const int t0i = BR * it + 2 * rp, cb = f0 + 32 * m + 4 * h;
        const float* i0 = a.acc_in + (((size_t)b * H + t0i) * W + cb) * COUT + n;
        const int rowstride = W * COUT;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int dcol = (i & 3) + 8 * (i >> 2);
          if (cb + dcol < W) {
            if (t0i < H) acc0[i] = i0[dcol * COUT];
            if (t0i + 1 < H) acc1[i] = i0[rowstride + dcol * COUT];
          }
        }
      }

#pragma unroll
      for (int i = 0; i < 4; ++i) {  // input row key q = BR*it + 2*rp + i  (input row t = q - 1)
        const int q2 = 2 * rp + i;
        const int ringrow = ((q2 >= BR) ? blk1 : blk0) * BR + (q2 & (BR - 1));
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int p = ringrow * SLOTS + 32 * m + r + dx;
          const int s = lds_swz<PB>(p);
          const char* base = smem + p * PB;
#pragma unroll
          for (int kg = 0; kg < NKG; ++kg) {
            const uint4 av = *(const uint4*)(base + (((2 * kg + h) ^ s) << 4));
            if (i <= 2) acc0 = Mma<T>::run(av, w[i * 3 + dx][kg], acc0);
            if (i >= 1) acc1 = Mma<T>::run(av, w[(i - 1) * 3 + dx][kg], acc1);
          }
        }
      }

      // ---- fused epilogue: + folded bias, ReLU, pool / time-sum.  Register i of the accumulator is pixel column
      // colbase + dcol(i), dcol(i) = (i&3) + 8*(i>>2); addresses are one per-lane base pointer + scalar offsets.
      const int t0 = BR * it + 2 * rp;  // pre-pool rows t0, t0+1
      const int colbase = f0 + 32 * m + 4 * h;
      if (EPI == EPI_POOL_H2) {
        const int Ho = H >> 1, to = t0 >> 1;
        if (to < Ho) {
          T* o0 = (T*)a.out + (((size_t)b * Ho + to) * W + colbase) * COUT + n;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int dcol = (i & 3) + 8 * (i >> 2);
            const float v = 0.5f * (fmaxf(acc0[i] + bv, 0.f) + fmaxf(acc1[i] + bv, 0.f));
            if (colbase + dcol < W) o0[dcol * COUT] = cvt_out<T>(v);
          }
        }
      } else if (EPI == EPI_POOL_2X2) {
        const int Ho = H >> 1, Wo = W >> 1, to = t0 >> 1;
        if (to < Ho) {
          T* o0 = (T*)a.out + (((size_t)b * Ho + to) * Wo + (colbase >> 1)) * COUT + n;
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const int dfo = ((i & 3) + 8 * (i >> 2)) >> 1;
            const float v = 0.25f * (fmaxf(acc0[i] + bv, 0.f) + fmaxf(acc0[i + 1] + bv, 0.f) +
                                     fmaxf(acc1[i] + bv, 0.f) + fmaxf(acc1[i + 1] + bv, 0.f));
            if ((colbase >> 1) + dfo < Wo) o0[dfo * COUT] = cvt_out<T>(v);
          }
        }
      } else if (EPI == EPI_MEAN_T) {
        const float k0 = (t0 < H) ? 1.f : 0.f, k1 = (t0 + 1 < H) ? 1.f : 0.f;
#pragma unroll
        for (int mm = 0; mm < MT; ++mm)
          if (mm == m) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
              cs[mm][i] += k0 * fmaxf(acc0[i] + bv, 0.f) + k1 * fmaxf(acc1[i] + bv, 0.f);
          }
      } else if (EPI == EPI_RAW) {
        float* o0 = a.raw_out + (((size_t)b * H + t0) * W + colbase) * COUT + n;
        const int rowstride = W * COUT;
        const bool r0ok = t0 < H, r1ok = t0 + 1 < H;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int dcol = (i & 3) + 8 * (i >> 2);
          if (colbase + dcol < W) {
            if (r0ok) o0[dcol * COUT] = acc0[i];
            if (r1ok) o0[rowstride + dcol * COUT] = acc1[i];
          }
        }
      } else {  // EPI_PLAIN
        T* o0 = (T*)a.out + (((size_t)b * H + t0) * W + colbase) * COUT + n;
        const int rowstride = W * COUT;
        const bool r0ok = t0 < H, r1ok = t0 + 1 < H;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int dcol = (i & 3) + 8 * (i >> 2);
          float v0 = acc0[i] + bv, v1 = acc1[i] + bv;
          if (a.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
          if (colbase + dcol < W) {
            if (r0ok) { o0[dcol * COUT] = cvt_out<T>(v0); st1 += v0; st2 = fmaf(v0, v0, st2); }
            if (r1ok) { o0[rowstride + dcol * COUT] = cvt_out<T>(v1); st1 += v1; st2 = fmaf(v1, v1, st2); }
          }
        }
      }
    }

    if (pf && !DMA) stage_store(blk2);
    __syncthreads();
    blk0 = blk1;
  }

  if (EPI == EPI_PLAIN) {
    if (a.stats_partial) {  // lanes r and r+32 hold the same channel; M-group waves too: combine through LDS
      st1 += __shfl_xor(st1, 32, 64);
      st2 += __shfl_xor(st2, 32, 64);
      float* red = (float*)smem;
      if (h == 0) { red[((mg * NSL + nsl) * 32 + r) * 2] = st1; red[((mg * NSL + nsl) * 32 + r) * 2 + 1] = st2; }
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
  }
  if (EPI == EPI_MEAN_T) {
    // column sums -> LDS [channel][column] (ring is free after the last barrier) -> coalesced rows of emb
    float* ef = (float*)smem;
    constexpr int LDW = C::EPI_LDW;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        ef[(nsl * 32 + r) * LDW + 32 * m + (i & 3) + 8 * (i >> 2) + 4 * h] = cs[m][i] * a.inv_h;
    __syncthreads();
    for (int e = tid; e < NSL * 32 * 32 * MT; e += NT) {
      const int nn = e / (32 * MT), col = e - nn * (32 * MT);
      const int f = f0 + col;
      if (f < W) a.emb[((size_t)b * COUT + cout_base + nn) * W + f] = ef[nn * LDW + col];
    }
  }
}

// host-side launcher (defined per instantiation in conv3x3_inst_*.hip)
template <typename T, int CIN, int NSL, int MG, int RP, int MT, int EPI, int MINW, bool ACCIN = false, bool DMA = false>
hipError_t launch_conv3x3(const ConvArgs& a0, hipStream_t stream) {
  using C = ConvCfg<T, CIN, NSL, MG, RP, MT, EPI>;
  ConvArgs a = a0;
  a.nstrips = (a.W + 32 * MT - 1) / (32 * MT);
  auto kern = conv3x3_mfma_kernel<T, CIN, NSL, MG, RP, MT, EPI, MINW, ACCIN, DMA>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(a.B * a.nstrips, a.COUT / (NSL * 32), 1), block(C::NT, 1, 1);
  hipLaunchKernelGGL(kern, grid, block, C::LDS_BYTES, stream, a);
  return hipGetLastError();
}

}  // namespace dfa
