// convt2x2_mfma.h -- ConvTranspose2d(kernel 2, stride 2) + BatchNorm(eval, folded) + ReLU on the matrix cores
// (src/model_cae.py:63-65, 68-71, 74-76).
//
// kernel == stride  =>  the transposed convolution does not overlap: every input pixel (i,j) produces its own 2x2
// output patch,   out[2i+a][2j+c][co] = bias[co] + sum_ci x[i][j][ci] * W[ci][co][a][c]
// i.e. one GEMM  [pixels x Cin] . [Cin x 4*Cout]  followed by a pixel shuffle (exact).
// GEMM column n = (2a+c)*Cout + co.  Input/output activations are channels-last, so the input tile of a workgroup
// (MSUB*32 consecutive pixels of the flattened [B*H*W] index) is ONE contiguous block of HBM -> LDS, no halo.
// A fragments come from LDS with the same chunk swizzle as conv3x3_mfma; each wave walks the 32-column N slices
// assigned to it, loading that slice's B fragments (Cin/KG x 16 bytes per lane) once per workgroup tile.
#pragma once
#include "conv3x3_mfma.h"

namespace dfa {

struct ConvTArgs {
  const void* in;      // [B][H][W][CIN] T
  const uint4* wpack;  // [4*COUT/32][CIN/KG][64] x 16 bytes
  const float* bias;   // [COUT] folded
  void* out;           // [B][2H][2W+opad_w][COUT] T
  int B, H, W, COUT, opad_w;
  int no_relu;         // 1: store bias + sum without the ReLU (train mode: BatchNorm runs as its own pass)
  // STATS form (train mode): per-workgroup [COUT][2] records of the sum / sum of squares of the values stored (fp32, before the
  // rounding for storage); workgroup 0 adds the opad_w column (bias only, written by cae_opad_col_kernel): B * 2H pixels of bias
  float* stats_partial;
};

__device__ __forceinline__ void convt_st4(bf16_t* p, const float* v) {
  *(uint2*)p = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
}
__device__ __forceinline__ void convt_st4(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }

template <typename T, int CIN, int MSUB, bool STATS = false>
__global__ __launch_bounds__(256) void convt2x2_mfma_kernel(ConvTArgs a) {
  constexpr int ES = sizeof(T), KG = 32 / ES, NKG = CIN / KG, PB = CIN * ES, CPP = PB / 16;
  constexpr int MTILE = 32 * MSUB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* obase = (int*)(smem + MTILE * PB);  // per tile pixel: output pixel index of (2i, 2j), or -1

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int H = a.H, W = a.W, COUT = a.COUT;
  const int Ho = 2 * H, Wo = 2 * W + a.opad_w;
  const long P = (long)a.B * H * W;
  const long g0 = (long)blockIdx.x * MTILE;

  // stage the contiguous input tile (zero beyond the last pixel)
  const char* src = (const char*)a.in + g0 * PB;
  for (int g = tid; g < MTILE * CPP; g += 256) {
    const int p = g / CPP, c = g % CPP;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (g0 + p < P) v = *(const uint4*)(src + (size_t)g * 16);
    *(uint4*)(smem + p * PB + ((c ^ lds_swz<(PB > 512 ? 512 : PB)>(p)) << 4)) = v;
  }
  for (int p = tid; p < MTILE; p += 256) {
    const long g = g0 + p;
    int v = -1;
    if (g < P) {
      const int bb = (int)(g / ((long)H * W));
      const int rem = (int)(g - (long)bb * H * W);
      const int ii = rem / W, jj = rem - ii * W;
      v = (bb * Ho + 2 * ii) * Wo + 2 * jj;
    }
    obase[p] = v;
  }
  __syncthreads();

  const int nslices = 4 * COUT / 32;
  T* out = (T*)a.out;
  // The WEIGHTS are the A operand (rows = 32 output channels of the slice), the pixels the columns: a lane owns ONE output pixel and
  // registers 4 g .. 4 g + 3 hold its channels 8 g + 4 h + (0..3) -- four 8-byte (bf16) / 16-byte (fp32) stores per lane and slice
  // instead of sixteen 2-byte ones (same products in the same k order: the stored values did not change).
  // STATS: slices wave, wave + 4, ... of one wave all carry the SAME 32 output channels (nslices = 4 COUT / 32 with COUT / 32 in
  // {1, 2, 4} groups: slice s <-> group s % (COUT / 32), and 4 is a multiple of that) at different patch positions: one running
  // pair per register
  float st1[STATS ? 16 : 1], st2[STATS ? 16 : 1];
  if (STATS) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { st1[i] = 0.f; st2[i] = 0.f; }
  }
  for (int s = wave; s < nslices; s += 4) {
    uint4 wb[NKG];
    const uint4* wp = a.wpack + (size_t)s * NKG * 64 + lane;
#pragma unroll
    for (int kg = 0; kg < NKG; ++kg) wb[kg] = wp[kg * 64];
    const int nglob = s * 32;            // all 32 channels of a slice share the patch position q (COUT % 32 == 0)
    const int q = nglob / COUT;
    const int cb = nglob - q * COUT;     // first channel of the slice
    const int oshift = (q >> 1) * Wo + (q & 1);
    float bv[16];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b4 = *(const float4*)(a.bias + cb + 8 * g + 4 * h);
      bv[4 * g] = b4.x; bv[4 * g + 1] = b4.y; bv[4 * g + 2] = b4.z; bv[4 * g + 3] = b4.w;
    }
#pragma unroll
    for (int ms = 0; ms < MSUB; ++ms) {
      f32x16_t acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.f;
      const int p = ms * 32 + r;
      const int sw = lds_swz<(PB > 512 ? 512 : PB)>(p);
      const char* base = smem + p * PB;
#pragma unroll
      for (int kg = 0; kg < NKG; ++kg) {
        const uint4 av = *(const uint4*)(base + (((2 * kg + h) ^ sw) << 4));
        acc = Mma<T>::run(wb[kg], av, acc);
      }
      const int ob = obase[p];
      if (ob >= 0) {
        T* o = out + (size_t)(ob + oshift) * COUT + cb + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[4 * g + e] + bv[4 * g + e];
            if (STATS) { st1[4 * g + e] += v[e]; st2[4 * g + e] = fmaf(v[e], v[e], st2[4 * g + e]); }
            if (!STATS && !a.no_relu) v[e] = fmaxf(v[e], 0.f);
          }
          convt_st4(o + 8 * g, v);
        }
      }
    }
  }
  if (STATS) {
    // register i of lane (r, h): channel (i & 3) + 8 (i >> 2) + 4 h of the wave's group, this lane's pixels: add the 32 pixel lanes of
    // each half, park one pair per (wave, channel), then add the waves of a group in wave order (a fixed order: reproducible records)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) {
        st1[i] += __shfl_xor(st1[i], off, 64);
        st2[i] += __shfl_xor(st2[i], off, 64);
      }
    }
    __syncthreads();                                   // every wave is done with the input tile: its first bytes become the scratch
    float* red = (float*)smem;
    if (r == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = (i & 3) + 8 * (i >> 2) + 4 * h;
        red[(wave * 32 + c) * 2] = st1[i];
        red[(wave * 32 + c) * 2 + 1] = st2[i];
      }
    }
    __syncthreads();
    if (tid < COUT) {
      const int ng = COUT / 32, g = tid >> 5, rr = tid & 31;
      float s1 = 0.f, s2 = 0.f;
      for (int w = g; w < 4; w += ng) { s1 += red[(w * 32 + rr) * 2]; s2 += red[(w * 32 + rr) * 2 + 1]; }
      if (a.opad_w && blockIdx.x == 0) {
        const float bvv = a.bias[tid], np = (float)(a.B * Ho);
        s1 = fmaf(np, bvv, s1);
        s2 = fmaf(np * bvv, bvv, s2);
      }
      float* dst = a.stats_partial + ((size_t)blockIdx.x * COUT + tid) * 2;
      dst[0] = s1;
      dst[1] = s2;
    }
  }
}

// records of the STATS form
template <int MSUB>
inline int convt2x2_blocks(long P) { return (int)((P + 32 * MSUB - 1) / (32 * MSUB)); }

template <typename T, int CIN, int MSUB, bool STATS = false>
hipError_t launch_convt2x2(const ConvTArgs& a, hipStream_t stream) {
  constexpr int PB = CIN * (int)sizeof(T), MTILE = 32 * MSUB;
  constexpr int LDS = MTILE * PB + MTILE * 4;
  if (STATS && (!a.stats_partial || !a.no_relu || a.COUT > 128 || (a.COUT != 32 && a.COUT != 64 && a.COUT != 128))) return hipErrorInvalidValue;
  auto kern = convt2x2_mfma_kernel<T, CIN, MSUB, STATS>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const long P = (long)a.B * a.H * a.W;
  dim3 grid((unsigned)((P + MTILE - 1) / MTILE)), block(256);
  hipLaunchKernelGGL(kern, grid, block, LDS, stream, a);
  return hipGetLastError();
}

}  // namespace dfa
