// convt_dgrad_bf16.hip -- data gradient of ConvTranspose2d(kernel 2, stride 2) in the auto-encoder's bf16 training step
// (autograd of src/model_cae.py:63-79 inside loss.backward(), src/train_cae.py:71) on the bf16 matrix cores.
//
// kernel == stride makes the layer a plain matrix product on the patch-major view of its output (gemm_f32.hip header):
//     dX[P x Cin] = Zp[P x K] . Wq^T,   K = 4 Cout,   Zp = pixel-unshuffled dz (bf16), Wq[Cin][K] = the layer's weight (fp32).
// Both operands have the reduction index contiguous, which is exactly what v_mfma_f32_32x32x16_bf16 fragments want (8
// consecutive k per lane): WEIGHTS are the A operand (32 input channels x 16 k), converted to bf16 fragments by a small pack
// kernel and held in registers for all of K (the forward of the same layer multiplies by the same bf16 weights, so this IS the gradient of the
// function the forward computed); pixels are the columns.  A workgroup streams tiles of PW x 32 consecutive pixels (one
// contiguous 32 KB block of Zp) through a double-buffered LDS image with the chunk swizzle of cae_dec_fused.hip -- the next
// tile's loads are in flight while the current one is multiplied -- and writes dX as bf16 directly.
//   dec1: K = 512, Cin = 256: wave = one 32-channel tile (128 weight VGPRs), 4 tiles per workgroup, grid.y = 2
//   dec2: K = 256, Cin = 128: wave = two tiles, two pixel streams per workgroup
//   dec3: K = 128, Cin = 64:  wave = both tiles, four pixel streams
// It replaces gemm_f32_kernel<bf16, float> (fp32 matrix cores, scalar staging: 0.33 ms per layer) + the fp32 -> bf16 cast pass.
#include "dfa_internal.h"
#include "conv3x3_mfma.h"

namespace dfa {

template <int K, int NPW>
__global__ __launch_bounds__(256, 2) void convt_dgrad_bf16_kernel(const bf16_t* __restrict__ zp, const uint4* __restrict__ wfrag,
                                                               bf16_t* __restrict__ dx, long P, int Cin, int waves_n, int ntiles_p) {
  constexpr int NKS = K / 16, ROWB = K * 2, CPR = K / 8;          // k-steps; bytes and 16-byte chunks per pixel row
  constexpr int TILE_B = 32 * 1024, NLD = TILE_B / (256 * 16);    // a tile = PW x 32 pixels = 32 KB for every instantiation: 8 DMA pieces per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, h = lane >> 5;
  const int PW = 4 / waves_n, rows_per_tile = 32 * PW;
  const int wn = wave % waves_n, ps = wave / waves_n;              // this wave's channel-tile group and pixel stream
  const int n0 = (blockIdx.y * waves_n + wn) * NPW * 32;           // first input channel of the wave's tiles

  // ---- weights: this wave's A fragments, packed bf16 by convt_dgrad_pack_kernel ([Cin / 32][NKS][64] x 16 bytes)
  uint4 w[NPW][NKS];
#pragma unroll
  for (int ni = 0; ni < NPW; ++ni)
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) w[ni][ks] = wfrag[((size_t)(n0 / 32 + ni) * NKS + ks) * 64 + lane];

  // tile -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers -- with 128 weight VGPRs they would spill): one wave
  // instruction fills 64 consecutive PHYSICAL chunks, the swizzle lives in the per-lane SOURCE address; rows beyond P are clamped
  auto tile_dma = [&](int t, int buf) {
    const long r0 = (long)t * rows_per_tile;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
      const int q = k * 256 + tid, p = q / CPR, c = (q % CPR) ^ (p & 15);
      const long row = min(r0 + p, P - 1);
      const char* src = (const char*)zp + (size_t)row * ROWB + (size_t)c * 16;
      char* dst = smem + buf * TILE_B + (k * 256 + wave * 64) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  int t = blockIdx.x, cur = 0;
  if (t < ntiles_p) tile_dma(t, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (; t < ntiles_p; t += gridDim.x) {
    const int tn = t + gridDim.x;
    if (tn < ntiles_p) tile_dma(tn, cur ^ 1);                      // in flight under this tile's MFMAs (the other buffer: every wave left it at the last barrier)
    const int prow = ps * 32 + col;
    const char* xb = smem + cur * TILE_B + prow * ROWB;
    const int sw = prow & 15;
    f32x16_t acc[NPW];
#pragma unroll
    for (int ni = 0; ni < NPW; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;
    // fragment reads four k-steps at a time (left alone, hipcc hoists all NKS reads to the top: 128 more VGPRs, one wave per SIMD)
#pragma unroll
    for (int k4 = 0; k4 < NKS; k4 += 4) {
      uint4 xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) xv[u] = *(const uint4*)(xb + (((2 * (k4 + u) + h) ^ sw) << 4));
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int ni = 0; ni < NPW; ++ni) acc[ni] = Mma<bf16_t>::run(w[ni][k4 + u], xv[u], acc[ni]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // dX[pixel][channel]: lane = pixel, registers 4 g .. 4 g + 3 = channels n0 + 32 ni + 8 g + 4 h + (0..3): 8-byte stores
    const long p = (long)t * rows_per_tile + prow;
    if (p < P) {
#pragma unroll
      for (int ni = 0; ni < NPW; ++ni) {
        bf16_t* o = dx + (size_t)p * Cin + n0 + 32 * ni + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(uint2*)(o + 8 * g) = make_uint2(pack_bf16x2(acc[ni][4 * g], acc[ni][4 * g + 1]), pack_bf16x2(acc[ni][4 * g + 2], acc[ni][4 * g + 3]));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
}

// Wq[Cin][K] fp32 -> bf16 (RNE, as the forward's packed image) A fragments: lane (row n = 32 tile + (lane & 31), half hh), element j
// <-> k = 16 ks + 8 hh + j
__global__ void convt_dgrad_pack_kernel(const float* __restrict__ wq, uint4* __restrict__ wfrag, int Cin, int K) {
  const int i = blockIdx.x * 256 + threadIdx.x, nks = K / 16;
  if (i >= (Cin / 32) * nks * 64) return;
  const int lane = i & 63, ks = (i >> 6) % nks, tile = (i >> 6) / nks;
  const float* wr = wq + (size_t)(32 * tile + (lane & 31)) * K + 16 * ks + 8 * (lane >> 5);
  const float4 a = *(const float4*)wr, b = *(const float4*)(wr + 4);
  wfrag[i] = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
}

bool convt_dgrad_bf16_supports(int Cin, int Cout) {
  return (Cin == 256 && Cout == 128) || (Cin == 128 && Cout == 64) || (Cin == 64 && Cout == 32);
}

// zp [P][4 Cout] bf16, wq [Cin][4 Cout] fp32, dx [P][Cin] bf16; wfrag = Cin * 4 Cout * 2 bytes of scratch for the bf16 fragments
hipError_t launch_convt_dgrad_bf16(const void* zp, const float* wq, void* wfrag, void* dx, long P, int Cin, int Cout, hipStream_t s) {
  if (!convt_dgrad_bf16_supports(Cin, Cout) || P < 1) return hipErrorInvalidValue;
  const int K = 4 * Cout, npw = (K == 512) ? 1 : 2;
  {
    const int n = (Cin / 32) * (K / 16) * 64;
    hipLaunchKernelGGL(convt_dgrad_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, s, wq, (uint4*)wfrag, Cin, K);
  }
  const int ntiles_n = Cin / 32, tiles_per_wg = ntiles_n < 4 * npw ? ntiles_n : 4 * npw;
  const int waves_n = tiles_per_wg / npw, PW = 4 / waves_n;
  const int ntiles_p = (int)((P + 32 * PW - 1) / (32 * PW));
  const int gx = ntiles_p < 1024 ? ntiles_p : 1024;
  dim3 grid(gx, ntiles_n / tiles_per_wg), block(256);
  const size_t lds = 2 * 32 * 1024;
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, block, lds, s, (const bf16_t*)zp, (const uint4*)wfrag, (bf16_t*)dx, P, Cin, waves_n, ntiles_p);
    return hipGetLastError();
  };
  if (K == 512) return go(convt_dgrad_bf16_kernel<512, 1>);
  if (K == 256) return go(convt_dgrad_bf16_kernel<256, 2>);
  return go(convt_dgrad_bf16_kernel<128, 2>);
}

}  // namespace dfa
