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
                                                               float* __restrict__ partial, int B, int H, int W) {
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
      if (f0 + p < W) widen8<T>(dz + ((((size_t)b * H + t) * W + f0 + p) * COUT + cg * 8), v);
      float4* d = reinterpret_cast<float4*>(dzs + p * COUT + cg * 8);
      d[0] = make_float4(v[0], v[1], v[2], v[3]);
      d[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
    for (int e = tid; e < 3 * (WG_SEG + 2) * (CIN / 8); e += 256) {
      const int cg = e % (CIN / 8);
      const int sl = (e / (CIN / 8)) % (WG_SEG + 2), row = e / ((CIN / 8) * (WG_SEG + 2));
      const int tt = t + row - 1, ff = f0 - 1 + sl;
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (tt >= 0 && tt < H && ff >= 0 && ff < W) widen8<T>(a + ((((size_t)b * H + tt) * W + ff) * CIN + cg * 8), v);
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

template <typename T, int CIN, int COUT>
static hipError_t launch_wgrad_t(const void* dz, const void* a, float* partial, int B, int H, int W, int nwg,
                                 hipStream_t s) {
  constexpr int LDS = (WG_SEG * COUT + 3 * (WG_SEG + 2) * CIN) * 4;
  auto kern = wgrad3x3_mfma_kernel<T, CIN, COUT>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), LDS, s, (const T*)dz, (const T*)a, partial, B, H, W);
  return hipGetLastError();
}

// dW [COUT][CIN][3][3] and db [COUT] <- dz [B][H][W][COUT], a [B][H][W][CIN]; partial: nwg * (COUT*CIN*9 + COUT) floats
hipError_t launch_wgrad3x3(int prec, int cin, int cout, const void* dz, const void* a, float* partial, float* dw,
                           float* db, int B, int H, int W, int nwg, hipStream_t s) {
  hipError_t e;
  if (prec == DFA_PREC_BF16) {
    if (cin == 64 && cout == 128) e = launch_wgrad_t<bf16_t, 64, 128>(dz, a, partial, B, H, W, nwg, s);
    else if (cin == 32 && cout == 64) e = launch_wgrad_t<bf16_t, 32, 64>(dz, a, partial, B, H, W, nwg, s);
    else return hipErrorInvalidValue;
  } else {
    if (cin == 64 && cout == 128) e = launch_wgrad_t<float, 64, 128>(dz, a, partial, B, H, W, nwg, s);
    else if (cin == 32 && cout == 64) e = launch_wgrad_t<float, 32, 64>(dz, a, partial, B, H, W, nwg, s);
    else return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  const int n = cout * cin * 9;
  // the weight block and the bias block of the partial records are reduced by two strided launches
  e = launch_reduce_partials_strided(partial, nwg, n + cout, 0, n, dw, s);
  if (e != hipSuccess) return e;
  return launch_reduce_partials_strided(partial, nwg, n + cout, n, cout, db, s);
}

}  // namespace dfa
