// gemm_tn_bf16.hip -- C[M][N] = sum_k X[k][m] * Z[k][n] for bf16 row-major X [K x M] and Z [K x N], fp32 accumulate on
// v_mfma_f32_32x32x16_bf16: the weight gradient of ConvTranspose2d(k2,s2) in the bf16 training mode of the auto-encoder,
// dWq[Cin][4Cout] = X^T . Zp with K = all input pixels of the layer (autograd of src/model_cae.py:63-79 inside
// loss.backward(), src/train_cae.py:71; gemm_f32.hip has the derivation).  The reduction index (pixel) is the SLOW index
// of both operands, so both MFMA operands are read from LDS with ds_read_b64_tr_b16 (8 consecutive k of one column), the
// recipe of wgrad_mfma.hip.  The three layers are 30 GFLOP each on 0.17-0.7 GB of input: HBM-bound, so the kernel keeps
// the tile small (64 x 128 per workgroup, 32 accumulator registers per lane, many workgroups per CU) and splits K over
// workgroups; partial[part][M][N] is reduced in a fixed order by the caller (launch_reduce_partials).
// Replaces two gemm_f32 launches per layer that ran the same product on the fp32 matrix cores at 19 TFLOP/s (1.6 ms).
#include "dfa_internal.h"

namespace dfa {

typedef __attribute__((ext_vector_type(4))) short tn_s16x4_t;

namespace tn {
constexpr int TM = 64, TN = 128, TK = 64;                 // workgroup tile and k-rows per staged item
constexpr int XS = TM * 2 + 64, ZS = TN * 2 + 64;         // LDS bytes per k-row: == 64 (mod 128), conflict-free tr reads
constexpr int X_BYTES = TK * XS, Z_BYTES = TK * ZS, BUF_BYTES = X_BYTES + Z_BYTES;
constexpr int NXC = TK * (TM / 8) / 256, NZC = TK * (TN / 8) / 256;   // 16-byte chunks per thread per item: 2, 4
}  // namespace tn

__global__ __launch_bounds__(256) void gemm_tn_bf16_kernel(const bf16_t* __restrict__ X, const bf16_t* __restrict__ Z,
                                                           float* __restrict__ partial, int M, int N, int K,
                                                           int items_per_part) {
  using namespace tn;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN, part = blockIdx.z;
  const int mt = wave & 1, np = wave >> 1;                  // wave tiles: (mt, 2*np) and (mt, 2*np + 1)
  const int i16 = lane & 15, qrow = i16 >> 2, pq = i16 & 3, chalf = (lane >> 4) & 1;
  const int x_lane = (8 * h + qrow) * XS + (mt * 32 + 16 * chalf + 4 * pq) * 2;
  const int z_lane = (8 * h + qrow) * ZS + (np * 64 + 16 * chalf + 4 * pq) * 2;

  f32x16_t acc[2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;

  const long item0 = (long)part * items_per_part;
  const long nitems_all = ((long)K + TK - 1) / TK;
  const long item1 = (item0 + items_per_part < nitems_all) ? item0 + items_per_part : nitems_all;
  uint4 sx[NXC], sz[NZC];
  auto load_item = [&](long item) {
    const long k0 = item * TK;
#pragma unroll
    for (int c = 0; c < NXC; ++c) {
      const int e = c * 256 + tid, kr = e / (TM / 8), cg = e % (TM / 8);
      const bool ok = k0 + kr < K;
      const uint4 v = *(const uint4*)(X + (ok ? (size_t)(k0 + kr) * M + m0 + cg * 8 : 0));
      sx[c] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int c = 0; c < NZC; ++c) {
      const int e = c * 256 + tid, kr = e / (TN / 8), cg = e % (TN / 8);
      const bool ok = k0 + kr < K;
      const uint4 v = *(const uint4*)(Z + (ok ? (size_t)(k0 + kr) * N + n0 + cg * 8 : 0));
      sz[c] = ok ? v : make_uint4(0u, 0u, 0u, 0u);
    }
  };
  auto store_item = [&](int buf) {
    char* xb = smem + buf * BUF_BYTES;
    char* zb = xb + X_BYTES;
#pragma unroll
    for (int c = 0; c < NXC; ++c) {
      const int e = c * 256 + tid, kr = e / (TM / 8), cg = e % (TM / 8);
      *(uint4*)(xb + kr * XS + cg * 16) = sx[c];
    }
#pragma unroll
    for (int c = 0; c < NZC; ++c) {
      const int e = c * 256 + tid, kr = e / (TN / 8), cg = e % (TN / 8);
      *(uint4*)(zb + kr * ZS + cg * 16) = sz[c];
    }
  };
  auto tr8 = [&](const char* p0, int stride4) {   // 8 consecutive k of the lane's column: two transposed reads
    const tn_s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4_t*)(p0));
    const tn_s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tn_s16x4_t*)(p0 + stride4));
    const uint2 u0 = __builtin_bit_cast(uint2, a0), u1 = __builtin_bit_cast(uint2, a1);
    return make_uint4(u0.x, u0.y, u1.x, u1.y);
  };

  long item = item0;
  if (item < item1) { load_item(item); store_item(0); }
  __syncthreads();
  for (int n = 0; item < item1; ++item, ++n) {
    if (item + 1 < item1) load_item(item + 1);
    const char* xb = smem + (n & 1) * BUF_BYTES;
    const char* zb = xb + X_BYTES;
#pragma unroll
    for (int ks = 0; ks < TK / 16; ++ks) {
      const uint4 xv = tr8(xb + ks * 16 * XS + x_lane, 4 * XS);          // B operand: columns = m
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint4 zv = tr8(zb + ks * 16 * ZS + z_lane + c * 64, 4 * ZS);  // A operand: rows = n
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, zv), __builtin_bit_cast(bf16x8_t, xv),
                                                         acc[c], 0, 0, 0);
      }
    }
    if (item + 1 < item1) store_item((n + 1) & 1);
    __syncthreads();
  }
  // acc[c][i]: n = n0 + (2*np + c)*32 + (i&3) + 8*(i>>2) + 4*h,  m = m0 + mt*32 + r
  float* out = partial + (size_t)part * M * N;
  // (registers 4 g .. 4 g + 3 are four consecutive n: one 16-byte store each)
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int nn = n0 + (2 * np + c) * 32 + 8 * g + 4 * h, mm = m0 + mt * 32 + r;
      *(float4*)(out + (size_t)mm * N + nn) = make_float4(acc[c][4 * g], acc[c][4 * g + 1], acc[c][4 * g + 2], acc[c][4 * g + 3]);
    }
}

// partial must hold *nparts * M * N floats; returns hipErrorInvalidValue for shapes the tile does not divide
hipError_t launch_gemm_tn_bf16(const void* X, const void* Z, float* partial, size_t partial_floats, int M, int N, int K,
                               int* nparts, hipStream_t s) {
  using namespace tn;
  if (M % TM || N % TN || K < 1) return hipErrorInvalidValue;
  const int nblk = (M / TM) * (N / TN);
  const long nitems = ((long)K + TK - 1) / TK;
  long parts = 768 / nblk > 0 ? 768 / nblk : 1;                       // ~3 workgroups per CU in flight
  if (parts > nitems) parts = nitems;
  while (parts > 1 && (size_t)parts * M * N > partial_floats) --parts;
  const int ipp = (int)((nitems + parts - 1) / parts);
  parts = (nitems + ipp - 1) / ipp;
  *nparts = (int)parts;
  hipLaunchKernelGGL(gemm_tn_bf16_kernel, dim3(N / TN, M / TM, (unsigned)parts), dim3(256), 0, s, (const bf16_t*)X,
                     (const bf16_t*)Z, partial, M, N, K, ipp);
  return hipGetLastError();
}

}  // namespace dfa
