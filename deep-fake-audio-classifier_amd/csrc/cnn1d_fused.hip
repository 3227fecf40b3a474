// cnn1d_fused.hip -- the whole CNN1D eval forward (src/model_cnn1d.py:37-46) as ONE kernel:
//   transpose (a view) -> 3 x [Conv1d(k3, pad 1) + BatchNorm1d (folded) + ReLU] -> AdaptiveAvgPool1d(1) -> Linear(128 -> 1).
//
// The path is bound by reading x once (231,124 B and 30.8 MFLOP per utterance); the three-launch form it replaces
// re-read x once per 32-channel output group, staged weights through LDS per 16-channel slab and kept fp32 intermediates in HBM
// (0.34 ms per 256 utterances, 2.8 % of the HBM roofline).  Here:
//   * workgroup = one utterance, 4 waves (one per SIMD); activations never leave the CU: h1 [32][T] and h2 [64][T] live in LDS
//     as fp32, channel-major with the frame index contiguous -- which is also the STORED feature layout [180][T], so layer 1
//     reads x in place through the caller's strides (any strides; consecutive lanes = consecutive frames = coalesced 128-byte
//     segments for the reference's [B,F,T] storage);
//   * every layer is an implicit GEMM on the fp32 matrix cores, v_mfma_f32_32x32x2_f32 (bit-exactly an fp32 fma chain: the 1e-4
//     logit bar holds with margin; no hi/lo splitting, no transposed staging): M = output channels (A operand = folded weights,
//     pre-packed in fragment order, read straight from L2), N = 32 frames, K = (tap, input channel) with a k-step = one tap x two
//     adjacent input channels, so the B operand of a lane is ONE float: x[ci + (lane >> 5)][t0 + (lane & 31) + tap - 1] -- a
//     conflict-free ds_read_b32 (layers 2, 3) or a coalesced global dword (layer 1, software-prefetched 12 k-steps ahead);
//   * layer 1: wave = frame tiles w, w+4, w+8 (one A fragment serves the wave's three tiles); layer 2: wave = (channel half,
//     frame-tile parity), its 48 A fragments in registers; layer 3: wave = 32 of the 128 channels for ALL frame tiles, its 96 A
//     fragments in registers, bias + ReLU + running sums over frames in the accumulator layout, one cross-lane reduction at the end;
//   * the classifier dot product (128 channels) is finished in the same kernel through 4 floats of LDS: logits[b] is the only
//     global write.
// MFMA budget per utterance at T = 321 (11 tiles): 3 x 276 + 6 x 48 + 11 x 96 = 2172 issue slots of 64 cycles on the busiest SIMD
// = 139 k cycles; one utterance per CU.  Shapes it takes: T <= 384 (activations must fit the 160 KB of LDS); anything else runs
// the three-launch path (api.hip).
#include "dfa_internal.h"
#include "conv3x3_mfma.h"

namespace dfa {
namespace c1f {
constexpr int TW = 32;        // frames per MFMA tile
constexpr int HALO = 4;       // an LDS row is [4 floats, col 3 = frame -1 = 0][32 * NT frames][4 floats, col 0 = frame 32 NT = 0]
constexpr int CH = 6;         // layer 1: channel pairs per software-pipeline chunk (24 twelve-byte loads in flight per wave)
constexpr int MAXT1 = 3;      // layer 1: tiles per wave (NT <= 12)
}  // namespace c1f

// A-fragment image of one folded Conv1d layer: wp[m][s][lane], m = 32-channel tile, s = 3 * cp + tap (cp = input-channel pair,
// zero-padded to ncp_pad pairs), lane: co = 32 m + (lane & 31), ci = 2 cp + (lane >> 5).
__global__ void pack_cnn1d_fused_kernel(const float* __restrict__ wf, float* __restrict__ wp, int cin, int cout, int ncp_pad, int tap_minor) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int total = (cout / 32) * ncp_pad * 3 * 64;
  if (i >= total) return;
  const int lane = i & 63;
  const int s = (i >> 6) % (ncp_pad * 3), m = (i >> 6) / (ncp_pad * 3);
  const int cp = s / 3, tap = s - 3 * cp;
  const int co = 32 * m + (lane & 31), ci = 2 * cp + (lane >> 5);
  // layers 2, 3: [m][s][lane] (one float per k-step and lane); layer 1 (tap_minor): [cp][lane][tap], so that the three tap
  // fragments of a channel pair are ONE 12-byte load per lane
  const size_t dst = tap_minor ? ((size_t)(m * ncp_pad + cp) * 64 + lane) * 3 + tap : (size_t)i;
  wp[dst] = (ci < cin) ? wf[((size_t)co * cin + ci) * 3 + tap] : 0.f;
}

// channel pairs of a layer's image: layer 1 (any cin) is padded to whole software-pipeline chunks; layers 2 and 3 (cin = 32, 64)
// are indexed with compile-time k-step counts (48, 96) and are not padded
int cnn1d_fused_ncp_pad(int cin, int layer) {
  const int ncp = (cin + 1) / 2;
  return layer == 0 ? (ncp + c1f::CH - 1) / c1f::CH * c1f::CH : ncp;
}
size_t cnn1d_fused_pack_floats(int cin, int cout, int layer) { return (size_t)(cout / 32) * cnn1d_fused_ncp_pad(cin, layer) * 3 * 64; }

hipError_t launch_pack_cnn1d_fused(const float* wf, float* wp, int cin, int cout, int layer, hipStream_t s) {
  const int total = (int)cnn1d_fused_pack_floats(cin, cout, layer);
  hipLaunchKernelGGL(pack_cnn1d_fused_kernel, dim3((total + 255) / 256), dim3(256), 0, s, wf, wp, cin, cout, cnn1d_fused_ncp_pad(cin, layer), layer == 0 ? 1 : 0);
  return hipGetLastError();
}

struct Cnn1dFusedArgs {
  const float* x;
  int64_t sb, st, sf;          // element (b, t, f) of x at b * sb + t * st + f * sf
  const float *wp1, *b1, *wp2, *b2, *wp3, *b3;   // packed folded weights / folded biases of the three layers
  const float *cw, *cb;        // classifier weight [128], bias [1]
  float* logits;               // [B]
  int T, F, NT, pitch, ncp1;   // NT = ceil(T / 32), pitch = 32 NT + 8 floats, ncp1 = padded channel pairs of layer 1
  long long* stamps;           // diagnostic (context option "clock_probe"): s_memtime at the phase boundaries of the first 128 workgroups
};

__device__ __forceinline__ f32x16_t mfma_f32(float a, float b, f32x16_t c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// acc = sum over NS k-steps of A[s] x (one LDS float per lane): k-step s = 3 cp + tap reads src[2 cp P + tap].  The reads run PD
// k-steps ahead of their MFMAs through a rotating window of registers, and `side(s)` -- the epilogue of the PREVIOUS tile, one
// accumulator register per k-step -- is issued in the shadow of the MFMAs; sched_barrier pins that order (left alone, hipcc
// issues each ds_read_b32 one or two MFMAs ahead of its use and puts the whole epilogue between two tiles: measured 1.1 - 1.5 k
// cycles per tile beside 3 - 6 k of MFMA time).
template <int NS, typename Side>
__device__ __forceinline__ void lds_gemm(const float (&A)[NS], const float* src, int P, f32x16_t& acc, Side side) {
  constexpr int PD = 12;
  float q[PD];
#pragma unroll
  for (int s = 0; s < PD; ++s) q[s] = src[2 * (s / 3) * P + s % 3];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const float bcur = q[s % PD];
    if (s + PD < NS) q[s % PD] = src[2 * ((s + PD) / 3) * P + (s + PD) % 3];
    acc = mfma_f32(A[s], bcur, acc);
    side(s);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Two frame tiles at once: the same A fragment feeds two independent accumulators, so a k-step is two back-to-back MFMAs that do
// not depend on each other (a single dependent chain of v_mfma_f32_32x32x2_f32 measured 78 - 88 cycles per MFMA here, not 64).
template <int NS, typename Side>
__device__ __forceinline__ void lds_gemm2(const float (&A)[NS], const float* src0, const float* src1, int P, f32x16_t& acc0,
                                          f32x16_t& acc1, Side side) {
  constexpr int PD = 8;
  float q0[PD], q1[PD];
#pragma unroll
  for (int s = 0; s < PD; ++s) {
    q0[s] = src0[2 * (s / 3) * P + s % 3];
    q1[s] = src1[2 * (s / 3) * P + s % 3];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const float b0 = q0[s % PD], b1 = q1[s % PD];
    if (s + PD < NS) {
      q0[s % PD] = src0[2 * ((s + PD) / 3) * P + (s + PD) % 3];
      q1[s % PD] = src1[2 * ((s + PD) / 3) * P + (s + PD) % 3];
    }
    acc0 = mfma_f32(A[s], b0, acc0);
    acc1 = mfma_f32(A[s], b1, acc1);
    side(s);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <bool SPAN>
__global__ __launch_bounds__(256) void cnn1d_fused_kernel(Cnn1dFusedArgs a) {
  using namespace c1f;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 31, hh = lane >> 5;
  const int b = blockIdx.x;
  const int T = a.T, NT = a.NT, P = a.pitch;
  float* const h1 = lds;                  // [32][P]
  float* const h2 = lds + 32 * P;         // [64][P]
  float* const red = lds + 96 * P;        // [4]

  const bool stamp = a.stamps != nullptr && tid == 0 && b < 128;
  if (stamp) { a.stamps[8 * b] = __builtin_amdgcn_s_memtime(); a.stamps[8 * b + 5] = __builtin_amdgcn_s_memrealtime(); }
  // weight fragments of layers 2 and 3 (48 + 96 registers): requested first, so their L2 / HBM latency hides under layer 1
  float A2[48], A3[96];
#pragma unroll
  for (int s = 0; s < 48; ++s) A2[s] = a.wp2[(size_t)((wave & 1) * 48 + s) * 64 + lane];
#pragma unroll
  for (int s = 0; s < 96; ++s) A3[s] = a.wp3[(size_t)(wave * 96 + s) * 64 + lane];

  // the halo columns (frame -1 and frames >= 32 NT) are zero for the whole kernel; the epilogues write every data column
  for (int i = tid; i < 96 * 8; i += 256) {
    const int r = i >> 3, c = i & 7;
    lds[r * P + (c < 4 ? c : 32 * NT + c)] = 0.f;
  }

  // ------------------------------------------------------------------------------------------------ layer 1: 180 -> 32
  {
    const float* xb = a.x + (int64_t)b * a.sb;
    const int ncp = a.ncp1;
    const int nmine = (NT - wave + 3) / 4;               // tiles wave, wave + 4, wave + 8
    f32x16_t acc[MAXT1];
#pragma unroll
    for (int j = 0; j < MAXT1; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // Per tile j the lane's frame is t = 32 (wave + 4 j) + col; tap k multiplies frame t - 1 + k (zero outside [0, T)).
    // SPAN (frames contiguous in memory, st == 1: the reference's stored [B,F,T] layout): the three taps of a lane are ONE
    // 12-byte load x[ts .. ts + 2], ts = clamp(t - 1, 0, T - 3); only the lanes at t = 0 (lo) and t = T - 1 (hi) see a clamped,
    // i.e. shifted, span and pick their values one position over.  Dword loads cost the CU's address unit ~20 cycles per
    // wave-instruction whatever their width: with one dword per MFMA (4 waves x 4 loads per 192 cycles) layer 1 ran at 348 cycles
    // per k-step against 192 of matrix-pipe time (stamps).  Generic strides keep one dword per (tap, tile).
    bool tin[MAXT1][3], lo[MAXT1], hi[MAXT1];
    unsigned toff[MAXT1][3];
#pragma unroll
    for (int j = 0; j < MAXT1; ++j) {
      const int t = TW * (wave + 4 * j) + col;
      const bool mine = (j < nmine) && t < T;
      lo[j] = mine && t == 0;
      hi[j] = mine && t == T - 1;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        tin[j][k] = mine && t - 1 + k >= 0 && t - 1 + k < T;
        toff[j][k] = SPAN ? (unsigned)min(max(t - 1, 0), T - 3) : (unsigned)min(max(t - 1 + k, 0), T - 1) * (unsigned)a.st;
      }
    }
    struct __attribute__((aligned(4))) F3 { float v[3]; };
    const F3* wl = reinterpret_cast<const F3*>(a.wp1) + lane;
    F3 A0[CH], A1[CH], B0[CH][MAXT1], B1[CH][MAXT1];
    // loads of channel pair cp: the three tap fragments of A, and per tile the lane's three x values
    auto load_cp = [&](int cp, F3& Aq, F3 (&Bq)[MAXT1]) {
      const unsigned cio = (unsigned)min(2 * cp + hh, a.F - 1) * (unsigned)a.sf;   // padded pairs: a valid address, the weight is zero
      Aq = wl[(size_t)cp * 64];
#pragma unroll
      for (int j = 0; j < MAXT1; ++j) {
        if (SPAN) {
          Bq[j] = *reinterpret_cast<const F3*>(xb + cio + toff[j][0]);
        } else {
#pragma unroll
          for (int k = 0; k < 3; ++k) Bq[j].v[k] = xb[cio + toff[j][k]];
        }
      }
    };
    auto bval = [&](const F3& q, int j, int k) -> float {
      if (!SPAN) return tin[j][k] ? q.v[k] : 0.f;
      if (k == 0) return hi[j] ? q.v[1] : (tin[j][0] ? q.v[0] : 0.f);
      if (k == 1) return lo[j] ? q.v[0] : (hi[j] ? q.v[2] : (tin[j][1] ? q.v[1] : 0.f));
      return lo[j] ? q.v[1] : (tin[j][2] ? q.v[2] : 0.f);
    };
    auto mfma_cp = [&](const F3& Aq, const F3 (&Bq)[MAXT1]) {       // 9 MFMAs; tiles past the wave's share multiply zeros
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < MAXT1; ++j) acc[j] = mfma_f32(Aq.v[k], bval(Bq[j], j, k), acc[j]);
    };
    // one chunk of MFMAs on (Ac, Bc) with the loads of chunk `cpn` into (An, Bn) spread between them, pair by pair: the wave
    // issues one instruction at a time, so a burst of loads between two chunks leaves the matrix pipe idle.
    auto chunk = [&](const F3 (&Ac)[CH], const F3 (&Bc)[CH][MAXT1], int cpn, F3 (&An)[CH], F3 (&Bn)[CH][MAXT1]) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        mfma_cp(Ac[c], Bc[c]);
        load_cp(cpn + c, An[c], Bn[c]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // No branch in the steady state: hipcc merges the vmcnt state of both sides of a branch conservatively, so a conditional
    // prefetch makes every wait cover the loads just issued (seen in the ISA: vmcnt(12) where vmcnt(59) was meant).  The chunk
    // index is clamped instead; a clamped reload is never consumed.
    const int nch = ncp / CH;
#pragma unroll
    for (int c = 0; c < CH; ++c) load_cp(c, A0[c], B0[c]);
    __builtin_amdgcn_sched_barrier(0);
    for (int c = 0; c + 1 < nch; c += 2) {
      chunk(A0, B0, CH * (c + 1), A1, B1);
      chunk(A1, B1, CH * min(c + 2, nch - 1), A0, B0);
    }
    if (nch & 1) {
#pragma unroll
      for (int c = 0; c < CH; ++c) mfma_cp(A0[c], B0[c]);
    }
    // bias + ReLU -> h1 (frames >= T are stored as zeros: they are layer 2's right-hand padding)
#pragma unroll
    for (int j = 0; j < MAXT1; ++j)
      if (j < nmine) {
        const int t = TW * (wave + 4 * j) + col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = (r & 3) + 8 * (r >> 2) + 4 * hh;
          const float v = fmaxf(acc[j][r] + a.b1[co], 0.f);
          h1[co * P + HALO + t] = t < T ? v : 0.f;
        }
      }
  }
  __syncthreads();
  if (stamp) a.stamps[8 * b + 1] = __builtin_amdgcn_s_memtime();

  // ------------------------------------------------------------------------------------------------ layer 2: 32 -> 64
  {
    const int m = wave & 1, par = wave >> 1;                 // 32 output channels, frame tiles par, par + 2, ...
    float bias[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias[r] = a.b2[32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh];
    // This wave's tiles are par, par + 2, ...; they are taken two at a time (lds_gemm2), and the bias + ReLU + store of a pair
    // runs in the shadow of the NEXT pair's MFMAs, one accumulator register per k-step.  A pair that does not exist (first trip)
    // stores to a scratch row instead of branching.
    float* const dummy = red + 8 + lane;
    auto store_reg = [&](const f32x16_t& v, int tile, int r, bool valid) {
      const int co = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh, t = TW * tile + col;
      float* dst = valid ? h2 + co * P + HALO + t : dummy;
      *dst = t < T ? fmaxf(v[r] + bias[r], 0.f) : 0.f;
    };
    auto store2 = [&](const f32x16_t& v0, const f32x16_t& v1, int tile0, bool valid, int s) {
      if (s >= 4 && s < 20) store_reg(v0, tile0, s - 4, valid);
      else if (s >= 20 && s < 36) store_reg(v1, tile0 + 2, s - 20, valid);
    };
    auto srcof = [&](int tile) { return h1 + hh * P + HALO + TW * tile + col - 1; };
    f32x16_t accA, accB, accC, accD;
#pragma unroll
    for (int r = 0; r < 16; ++r) accC[r] = accD[r] = 0.f;
    const int n2 = (NT - par + 1) / 2, npair = n2 / 2;       // tiles of this wave, whole pairs among them
    int pend = 0;                                            // which pair still owes its epilogue: 1 = (A, B), 2 = (C, D)
    int tp = par;                                            // first tile of the pending pair
    for (int p = 0; p < npair; p += 2) {
      const int t0 = par + 4 * p;
      lds_gemm2<48>(A2, srcof(t0), srcof(t0 + 2), P, accA, accB, [&](int s) { store2(accC, accD, t0 - 4, p > 0, s); });
      pend = 1; tp = t0;
      if (p + 1 < npair) {
        lds_gemm2<48>(A2, srcof(t0 + 4), srcof(t0 + 6), P, accC, accD, [&](int s) { store2(accA, accB, t0, true, s); });
        pend = 2; tp = t0 + 4;
      }
    }
    if (n2 & 1) {                                            // a last single tile; the pending pair's epilogue rides on it
      const int tl = par + 2 * (n2 - 1);
      if (pend == 2) {
        lds_gemm<48>(A2, srcof(tl), P, accA, [&](int s) { store2(accC, accD, tp, true, s); });
#pragma unroll
        for (int r = 0; r < 16; ++r) store_reg(accA, tl, r, true);
      } else {
        lds_gemm<48>(A2, srcof(tl), P, accC, [&](int s) { store2(accA, accB, tp, pend == 1, s); });
#pragma unroll
        for (int r = 0; r < 16; ++r) store_reg(accC, tl, r, true);
      }
    } else if (pend == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { store_reg(accA, tp, r, true); store_reg(accB, tp + 2, r, true); }
    } else if (pend == 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { store_reg(accC, tp, r, true); store_reg(accD, tp + 2, r, true); }
    }
  }
  __syncthreads();
  if (stamp) a.stamps[8 * b + 2] = __builtin_amdgcn_s_memtime();

  // ------------------------------------------------------------------------------------------------ layer 3: 64 -> 128, mean over frames, classifier
  {
    const int m = wave;                                      // 32 of the 128 output channels, every frame tile
    float bias[16], sum[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      bias[r] = a.b3[32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh];
      sum[r] = 0.f;
    }
    auto add_reg = [&](const f32x16_t& v, int tile, int r, bool valid) {
      sum[r] += (valid && TW * tile + col < T) ? fmaxf(v[r] + bias[r], 0.f) : 0.f;
    };
    auto add2 = [&](const f32x16_t& v0, const f32x16_t& v1, int tile0, bool valid, int s) {
      if (s >= 8 && s < 24) add_reg(v0, tile0, s - 8, valid);
      else if (s >= 24 && s < 40) add_reg(v1, tile0 + 1, s - 24, valid);
    };
    auto srcof = [&](int tile) { return h2 + hh * P + HALO + TW * tile + col - 1; };
    f32x16_t accA, accB, accC, accD;
#pragma unroll
    for (int r = 0; r < 16; ++r) accC[r] = accD[r] = 0.f;
    const int npair = NT / 2;
    int pend = 0, tp = 0;                                    // pending pair: 1 = (A, B), 2 = (C, D); its first tile
    for (int p = 0; p < npair; p += 2) {
      lds_gemm2<96>(A3, srcof(2 * p), srcof(2 * p + 1), P, accA, accB, [&](int s) { add2(accC, accD, 2 * p - 2, p > 0, s); });
      pend = 1; tp = 2 * p;
      if (p + 1 < npair) {
        lds_gemm2<96>(A3, srcof(2 * p + 2), srcof(2 * p + 3), P, accC, accD, [&](int s) { add2(accA, accB, 2 * p, true, s); });
        pend = 2; tp = 2 * p + 2;
      }
    }
    if (NT & 1) {
      if (pend == 2) {
        lds_gemm<96>(A3, srcof(NT - 1), P, accA, [&](int s) { add2(accC, accD, tp, true, s); });
#pragma unroll
        for (int r = 0; r < 16; ++r) add_reg(accA, NT - 1, r, true);
      } else {
        lds_gemm<96>(A3, srcof(NT - 1), P, accC, [&](int s) { add2(accA, accB, tp, pend == 1, s); });
#pragma unroll
        for (int r = 0; r < 16; ++r) add_reg(accC, NT - 1, r, true);
      }
    } else if (pend == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { add_reg(accA, tp, r, true); add_reg(accB, tp + 1, r, true); }
    } else if (pend == 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { add_reg(accC, tp, r, true); add_reg(accD, tp + 1, r, true); }
    }
    // sum over the 32 frame lanes of each half-wave (a half holds 16 channels), then the classifier's share of this wave
    float part = 0.f;
    const float inv_t = 1.0f / (float)T;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s = sum[r];
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
      part = fmaf(s * inv_t, a.cw[32 * m + (r & 3) + 8 * (r >> 2) + 4 * hh], part);
    }
    part += __shfl_xor(part, 32, 64);
    if (lane == 0) red[wave] = part;
  }
  __syncthreads();
  if (stamp) { a.stamps[8 * b + 3] = __builtin_amdgcn_s_memtime(); a.stamps[8 * b + 6] = __builtin_amdgcn_s_memrealtime(); }
  if (tid == 0) a.logits[b] = ((red[0] + red[1]) + (red[2] + red[3])) + a.cb[0];
}

size_t cnn1d_fused_lds_bytes(int T) {
  const int NT = (T + c1f::TW - 1) / c1f::TW;
  return ((size_t)96 * (32 * NT + 8) + 8 + 64) * sizeof(float);   // h1, h2, red[4] (+ pad), one scratch row
}
bool cnn1d_fused_supports(int T) {   // layer 1 gives a wave at most MAXT1 tiles (NT <= 12, T <= 384); the activations must fit LDS
  return T >= 3 && (T + c1f::TW - 1) / c1f::TW <= 4 * c1f::MAXT1 && cnn1d_fused_lds_bytes(T) <= 160 * 1024;
}

hipError_t launch_cnn1d_fused(const float* x, int64_t sb, int64_t st, int64_t sf, const float* wp1, const float* b1, const float* wp2,
                              const float* b2, const float* wp3, const float* b3, const float* cw, const float* cb, float* logits, int B,
                              int T, int F, hipStream_t s, long long* stamps) {
  Cnn1dFusedArgs a{};
  a.stamps = stamps;
  a.x = x; a.sb = sb; a.st = st; a.sf = sf;
  a.wp1 = wp1; a.b1 = b1; a.wp2 = wp2; a.b2 = b2; a.wp3 = wp3; a.b3 = b3; a.cw = cw; a.cb = cb; a.logits = logits;
  a.T = T; a.F = F; a.NT = (T + c1f::TW - 1) / c1f::TW; a.pitch = 32 * a.NT + 8; a.ncp1 = cnn1d_fused_ncp_pad(F, 0);
  const size_t lds = cnn1d_fused_lds_bytes(T);
  auto kern = (st == 1) ? cnn1d_fused_kernel<true> : cnn1d_fused_kernel<false>;
  hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, s, a);
  return hipGetLastError();
}

}  // namespace dfa
