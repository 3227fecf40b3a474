// train_elem.hip -- train-mode passes of the 2-D CNN that are not convolutions (src/train.py:71-76 through
// src/model.py:16-19,22-25,28-29,37-39 and torch autograd): BatchNorm with batch statistics (forward + backward),
// ReLU, AvgPool2d((2,1)), Dropout, mean over T, Linear backward, BCE-with-logits on smoothed labels, AdamW.
// All tensors are channels-last [B][H][W][C]; every pass is HBM-bound and moves 8 channels (16-32 bytes) per lane.
// Reductions are two-stage with a fixed order (per-block partials in fp32, final sum in fp64): deterministic.
#include "dfa_internal.h"
#include "rng.h"

namespace dfa {

template <typename T>
__device__ __forceinline__ void ld8(const T* p, float* v);
template <>
__device__ __forceinline__ void ld8<float>(const float* p, float* v) {
  const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <>
__device__ __forceinline__ void ld8<bf16_t>(const bf16_t* p, float* v) {
  const uint4 q = *reinterpret_cast<const uint4*>(p);
  const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(u[e] << 16); v[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
}
template <typename T>
__device__ __forceinline__ void st8(T* p, const float* v);
template <>
__device__ __forceinline__ void st8<float>(float* p, const float* v) {
  reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
template <>
__device__ __forceinline__ void st8<bf16_t>(bf16_t* p, const float* v) {
  bf16_t o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = float_to_bf16(v[j]);
  *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(o);
}

// ---- BatchNorm statistics: partial[k][C][2] (sum, sum of squares) -> mean, biased var, invstd; running stats update
// (torch.nn.BatchNorm2d train mode: running = (1-m)*running + m*batch, with the UNBIASED batch variance).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int nparts, int C, double n,
                                                          float* __restrict__ mean, float* __restrict__ var,
                                                          float* __restrict__ invstd, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float momentum) {
  __shared__ double r1[256], r2[256];
  const int c = blockIdx.x, tid = threadIdx.x;
  double s1 = 0.0, s2 = 0.0;
  for (int k = tid; k < nparts; k += 256) {
    s1 += (double)partial[((size_t)k * C + c) * 2];
    s2 += (double)partial[((size_t)k * C + c) * 2 + 1];
  }
  r1[tid] = s1; r2[tid] = s2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { r1[tid] += r1[tid + off]; r2[tid] += r2[tid + off]; }
    __syncthreads();
  }
  if (tid == 0) {
    const double m = r1[0] / n;
    double v = r2[0] / n - m * m;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)m;
    var[c] = (float)v;
    invstd[c] = (float)(1.0 / sqrt(v + (double)kBnEps));
    if (running_mean) {
      const double vu = (n > 1.0) ? v * n / (n - 1.0) : v;
      running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * m);
      running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * vu);
    }
  }
}

// generic fixed-order reduction of partial[k][n] over k (fp64 accumulate), out[j] = scale * sum.
// Block = 32 outputs x 8 k-lanes (coalesced 128-byte rows); grid.y splits k into chunks whose sums go to a second
// level (stage2 != nullptr) that a final launch of the same kernel adds up: always the same order -> deterministic.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, int nparts, int n,
                                                              float scale, float* __restrict__ out, int chunk) {
  __shared__ double red[8][33];
  const int jl = threadIdx.x & 31, kl = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + jl;
  const int k0 = blockIdx.y * chunk, k1 = min(nparts, k0 + chunk);
  double s = 0.0;
  if (j < n)
    for (int k = k0 + kl; k < k1; k += 8) s += (double)partial[(size_t)k * n + j];
  red[kl][jl] = s;
  __syncthreads();
  if (kl == 0 && j < n) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += red[q][jl];
    out[(size_t)blockIdx.y * n + j] = (float)(t * (double)scale);
  }
}

// out[j] = sum_k partial[k*stride + off + j]
__global__ void reduce_partials_strided_kernel(const float* __restrict__ partial, int nparts, int stride, int off, int n,
                                               float* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double s = 0.0;
  for (int k = 0; k < nparts; ++k) s += (double)partial[(size_t)k * stride + off + j];
  out[j] = (float)s;
}

// weight-gradient window: out[(co_off+co)][ci_off+ci][tap] = sum_k partial[k*stride + (co*cin + ci)*9 + tap]
__global__ void reduce_wgrad_window_kernel(const float* __restrict__ partial, int nparts, int stride, int cin, int cout,
                                           int cin_total, int ci_off, int co_off, float* __restrict__ dw) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cout * cin * 9) return;
  double s = 0.0;
  for (int k = 0; k < nparts; ++k) s += (double)partial[(size_t)k * stride + j];
  const int tap = j % 9, ci = (j / 9) % cin, co = j / (9 * cin);
  dw[((size_t)(co_off + co) * cin_total + ci_off + ci) * 9 + tap] = (float)s;
}

// whole weight-gradient record in one launch: elements [0, cout*cin*9) -> the dw window, [cout*cin*9, +cout) -> db.
// 64 elements x 4 groups of partial records per block; fp64 sums, combined in a fixed order.
__global__ __launch_bounds__(256) void reduce_wgrad_record_kernel(const float* __restrict__ partial, int nparts, int stride,
                                                                  int cin, int cout, int cin_total, int ci_off, int co_off,
                                                                  float* __restrict__ dw, float* __restrict__ db, int perm) {
  __shared__ double red[4][64];
  const int lane = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int n = cout * cin * 9, total = n + cout;
  const int e = blockIdx.x * 64 + lane;
  double s = 0.0;
  if (e < total) {
    const int k0 = (int)((long)pg * nparts / 4), k1 = (int)((long)(pg + 1) * nparts / 4);
    const float* p = partial + (size_t)k0 * stride + e;
#pragma unroll 8
    for (int k = k0; k < k1; ++k, p += stride) s += (double)*p;
  }
  red[pg][lane] = s;
  __syncthreads();
  if (pg != 0 || e >= total) return;
  const float v = (float)((red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
  if (e < n) {
    int tap, ci, co;
    if (perm) {   // accumulator-order record of wgrad3x3_bf16_v3_kernel: [tap][g][wave = is + IS cs][lane = 32 h + r][e4]
      const int IS = cin >> 5, nwr = IS * (cout >> 5);
      const int e4 = e & 3, ln = (e >> 2) & 63, w = (e >> 8) % nwr, tg = (e >> 8) / nwr;
      tap = tg >> 2;
      co = (w / IS) * 32 + e4 + 8 * (tg & 3) + 4 * (ln >> 5);
      ci = (w % IS) * 32 + (ln & 31);
    } else {
      tap = e % 9; ci = (e / 9) % cin; co = e / (9 * cin);
    }
    dw[((size_t)(co_off + co) * cin_total + ci_off + ci) * 9 + tap] = v;
  } else if (db) {
    db[co_off + e - n] = v;
  }
}

// ---- forward: BN(batch stats) + ReLU + AvgPool2d((2,1)) + Dropout        z[B][H][W][C] -> a[B][H/2][W][C]
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_poolh2_drop_kernel(const T* __restrict__ z, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, T* __restrict__ out,
                                                                  int B, int H, int W, int C, DropCfg dc) {
  const int Ho = H >> 1, CG = C >> 3;
  const size_t total = (size_t)B * Ho * W * CG;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cg = (int)(i % CG);
  const size_t pix = i / CG;                 // (b*Ho + to)*W + f
  const int f = (int)(pix % W);
  const size_t bt = pix / W;
  const int to = (int)(bt % Ho), b = (int)(bt / Ho);
  float z0[8], z1[8], o[8], ds[8];
  ld8<T>(z + ((((size_t)b * H + 2 * to) * W + f) * C + cg * 8), z0);
  ld8<T>(z + ((((size_t)b * H + 2 * to + 1) * W + f) * C + cg * 8), z1);
  drop_scale8(dc, pix * C + cg * 8, ds);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    const float sc = gamma[c] * invstd[c], sh = beta[c] - mean[c] * sc;
    o[j] = 0.5f * (fmaxf(fmaf(z0[j], sc, sh), 0.f) + fmaxf(fmaf(z1[j], sc, sh), 0.f)) * ds[j];
  }
  st8<T>(out + pix * C + cg * 8, o);
}

// ---- per-channel sum / sum of squares of a channels-last tensor z[npix][C] -> partial[block][C][2]
template <typename T>
__global__ __launch_bounds__(256) void cl_stats_kernel(const T* __restrict__ z, float* __restrict__ partial, size_t npix,
                                                       int C, int pix_per_block) {
  extern __shared__ float red[];  // [PL][C][2]
  const int CG = C >> 3, PL = 256 / CG;
  const int tid = threadIdx.x, cg = tid % CG, pl = tid / CG;
  const size_t p0 = (size_t)blockIdx.x * pix_per_block;
  const size_t p1 = (p0 + pix_per_block < npix) ? p0 + pix_per_block : npix;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  if (pl < PL) {
    // four pixels per trip: four independent 16-byte loads in flight per lane (one at a time left this pass at 0.8 TB/s: a
    // 4096-pixel block per workgroup gave 55-900 workgroups of serial single loads)
    size_t p = p0 + pl;
    for (; p + 3 * (size_t)PL < p1; p += 4 * (size_t)PL) {
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) ld8<T>(z + (p + u * (size_t)PL) * C + cg * 8, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s1[j] += v[u][j]; s2[j] = fmaf(v[u][j], v[u][j], s2[j]); }
    }
    for (; p < p1; p += PL) {
      float v[8];
      ld8<T>(z + p * C + cg * 8, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { s1[j] += v[j]; s2[j] = fmaf(v[j], v[j], s2[j]); }
    }
  }
  if (pl < PL) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[(pl * C + cg * 8 + j) * 2] = s1[j]; red[(pl * C + cg * 8 + j) * 2 + 1] = s2[j]; }
  }
  __syncthreads();
  for (int e = tid; e < C * 2; e += 256) {
    float s = 0.f;
    for (int q = 0; q < PL; ++q) s += red[q * C * 2 + e];
    partial[(size_t)blockIdx.x * C * 2 + e] = s;
  }
}

// ---- forward: BN(batch stats) + ReLU [+ AvgPool2d(2)]     z[B][H][W][C] -> out[B][H/P][W/P][C], P = 1 or 2
template <typename T, int P>
__global__ __launch_bounds__(256) void bn_relu_pool_kernel(const T* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, T* __restrict__ out, int B,
                                                           int H, int W, int C) {
  const int Ho = H / P, Wo = W / P, CG = C >> 3;
  const size_t total = (size_t)B * Ho * Wo * CG;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cg = (int)(i % CG);
  const size_t pix = i / CG;
  const int fo = (int)(pix % Wo);
  const size_t bt = pix / Wo;
  const int to = (int)(bt % Ho), b = (int)(bt / Ho);
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
  for (int a = 0; a < P; ++a)
#pragma unroll
    for (int c2 = 0; c2 < P; ++c2) {
      float v[8];
      ld8<T>(z + ((((size_t)b * H + P * to + a) * W + P * fo + c2) * C + cg * 8), v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        const float sc = gamma[c] * invstd[c], sh = beta[c] - mean[c] * sc;
        o[j] += fmaxf(fmaf(v[j], sc, sh), 0.f);
      }
    }
  if (P == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] *= 0.25f;
  }
  st8<T>(out + pix * C + cg * 8, o);
}

// ---- forward: BN(batch stats) + ReLU + mean over T       z[B][H][W][C] -> emb[B][C][W]
template <typename T, bool MSUM>
__global__ __launch_bounds__(256) void bn_relu_meant_kernel(const T* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ emb,
                                                            float* __restrict__ msum, int B, int H, int W, int C) {
  // MSUM (a template parameter: as a run-time test inside the t loop it cut the unrolled body into basic blocks and hipcc kept ONE
  // load in flight per thread -- 3.3 TB/s): also save, per (b, f, c), the two sums over t that the BatchNorm BACKWARD reduction of this layer needs --
  // n = sum_t mask, sx = sum_t mask * xhat (mask and xhat with the backward pass's own formulas) -- as msum[b][f][c][2].
  // With them dbeta = sum_{b,f} g * n and dgamma = sum_{b,f} g * sx (g = demb / H does not depend on t), so the backward
  // never re-reads z for its reduction (bn_bwd_reduce_saved_kernel below).
  const int CG = C >> 3;
  const size_t total = (size_t)B * W * CG;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cg = (int)(i % CG);
  const int f = (int)((i / CG) % W), b = (int)(i / ((size_t)CG * W));
  float sc[8], sh[8], acc[8], mu[8], is[8], gm[8], bt[8], cn[8], sx[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    mu[j] = mean[c]; is[j] = invstd[c]; gm[j] = gamma[c]; bt[j] = beta[c];
    sc[j] = gm[j] * is[j];
    sh[j] = bt[j] - mu[j] * sc[j];
    acc[j] = 0.f; cn[j] = 0.f; sx[j] = 0.f;
  }
#pragma unroll 4      // four 16-byte loads in flight per thread (the sums stay in t order)
  for (int t = 0; t < H; ++t) {
    float v[8];
    ld8<T>(z + ((((size_t)b * H + t) * W + f) * C + cg * 8), v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc[j] += fmaxf(fmaf(v[j], sc[j], sh[j]), 0.f);
      if (MSUM) {
        const float xh = (v[j] - mu[j]) * is[j];
        const bool on = fmaf(gm[j], xh, bt[j]) > 0.f;          // bn_bwd_apply_kernel's mask, bit for bit
        cn[j] += on ? 1.f : 0.f;
        sx[j] += on ? xh : 0.f;
      }
    }
  }
  const float inv_h = 1.0f / (float)H;
#pragma unroll
  for (int j = 0; j < 8; ++j) emb[((size_t)b * C + cg * 8 + j) * W + f] = acc[j] * inv_h;
  if (MSUM) {
    float* o = msum + (((size_t)b * W + f) * C + cg * 8) * 2;
#pragma unroll
    for (int j = 0; j < 8; j += 2) *reinterpret_cast<float4*>(o + 2 * j) = make_float4(cn[j], sx[j], cn[j + 1], sx[j + 1]);
  }
}

// BatchNorm backward reduction of the mean-over-T layer from the forward's saved sums: partial[block][C][2] with
// S1 = sum g*n, S2 = sum g*sx over this block's (b, f) positions, g = demb[b][f][c] / H.  Reads 3 floats per (b, f, c)
// instead of z [B][H][W][C].
__global__ __launch_bounds__(256) void bn_bwd_reduce_saved_kernel(const float* __restrict__ demb, const float* __restrict__ msum,
                                                                  float* __restrict__ partial, size_t npos, int C, float inv_h,
                                                                  int pos_per_block) {
  extern __shared__ float red[];  // [PL][C][2]
  const int CG = C >> 3, PL = 256 / CG;
  const int tid = threadIdx.x, cg = tid % CG, pl = tid / CG;
  const size_t p0 = (size_t)blockIdx.x * pos_per_block;
  const size_t p1 = (p0 + pos_per_block < npos) ? p0 + pos_per_block : npos;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
  if (pl < PL)
    for (size_t p = p0 + pl; p < p1; p += PL) {
      float d[8];
      ld8<float>(demb + p * C + cg * 8, d);
      const float* m = msum + (p * C + cg * 8) * 2;
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        const float4 q = *reinterpret_cast<const float4*>(m + 2 * j);
        const float g0 = d[j] * inv_h, g1 = d[j + 1] * inv_h;
        s1[j] = fmaf(g0, q.x, s1[j]); s2[j] = fmaf(g0, q.y, s2[j]);
        s1[j + 1] = fmaf(g1, q.z, s1[j + 1]); s2[j + 1] = fmaf(g1, q.w, s2[j + 1]);
      }
    }
  if (pl < PL) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[(pl * C + cg * 8 + j) * 2] = s1[j]; red[(pl * C + cg * 8 + j) * 2 + 1] = s2[j]; }
  }
  __syncthreads();
  for (int e = tid; e < C * 2; e += 256) {
    float s = 0.f;
    for (int q = 0; q < PL; ++q) s += red[q * C * 2 + e];
    partial[(size_t)blockIdx.x * C * 2 + e] = s;
  }
}

// ---- backward of Linear(K,1): demb[b][j] = dlogit[b]*w[j];  dw[j] = sum_b dlogit[b]*emb[b][j];  db = sum_b dlogit[b]
// tc > 0: demb is written TRANSPOSED per utterance, j = c*tw + f  ->  demb[b][f][c] (tc channels, tw columns), the
// channels-last order the BatchNorm backward of block 3 reads 8 channels at a time.
__global__ __launch_bounds__(256) void linear_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ w,
                                                         const float* __restrict__ emb, float* __restrict__ demb,
                                                         float* __restrict__ dw, float* __restrict__ db, int B, int K, int tc,
                                                         int tw) {
  // block = 32 columns j x 8 utterance groups: a thread walks B/8 utterances (one thread per column walked all B with a
  // dependent load->fma chain: 86 us at B = 256); the 8 partial dw sums meet in LDS in a fixed order
  __shared__ float red[8][32];
  const int jl = threadIdx.x & 31, bg = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + jl;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dlogits[b];
    db[0] = s;
  }
  float s = 0.f;
  if (j < K) {
    // the thread writes demb at OUTPUT position j (consecutive lanes, consecutive addresses) and fetches the weight that belongs
    // there once: position j = f * tc + c holds w[c * tw + f].  (Writing input column j's value at its transposed place put every
    // lane's 4 bytes 4 * tc bytes from its neighbour's: 23.6 MB of single-word stores, 62 us at B = 256.)
    const float wt = (tc > 0) ? w[(j % tc) * tw + j / tc] : w[j];
#pragma unroll 4
    for (int b = bg; b < B; b += 8) {
      const float d = dlogits[b];
      demb[(size_t)b * K + j] = d * wt;
      s = fmaf(d, emb[(size_t)b * K + j], s);
    }
  }
  red[bg][jl] = s;
  __syncthreads();
  if (bg == 0 && j < K)
    dw[j] = ((red[0][jl] + red[1][jl]) + (red[2][jl] + red[3][jl])) + ((red[4][jl] + red[5][jl]) + (red[6][jl] + red[7][jl]));
}

// ---- BatchNorm backward.  Upstream gradient dy of the BN *output* y = gamma*xhat+beta after the ReLU mask:
//   SRC_MEANT: dy = (y > 0) * demb[b][c][f] / H                                   (block 3: mean over T then Linear)
//   SRC_POOL : dy = (y > 0) * 0.5 * dropscale * da[b][t/2][f][c], rows t >= 2*(H/2) get 0   (blocks 1, 2)
// reduce: S1[c] = sum dy, S2[c] = sum dy*xhat  (= dbeta, dgamma);  apply: dz = gamma*invstd*(dy - S1/N - xhat*S2/N).
//   SRC_DIRECT: dy = (y > 0) * da[b][t][f][c]                                   (CAE decoder: ReLU feeds the next layer)
//   SRC_POOL22: dy = (y > 0) * 0.25 * da[b][t/2][f/2][c], rows/cols beyond 2*(H/2), 2*(W/2) get 0   (CAE encoder)
enum { SRC_MEANT = 0, SRC_POOL = 1, SRC_DIRECT = 2, SRC_POOL22 = 3 };

template <typename T, int SRC>
__device__ __forceinline__ void upstream8(const float* demb, const T* da, const DropCfg& dc, int b, int t, int f, int cg,
                                          int H, int W, int C, float inv_h, float* g) {
  if (SRC == SRC_MEANT) {
    float d[8];
    ld8<float>(demb + ((size_t)b * W + f) * C + cg * 8, d);   // demb is [B][W][C] (linear_bwd_kernel with tc > 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = d[j] * inv_h;
  } else if (SRC == SRC_DIRECT) {
    ld8<T>(da + (((size_t)b * H + t) * W + f) * C + cg * 8, g);
  } else if (SRC == SRC_POOL22) {
    // a last odd row / column was not pooled: zero gradient.  The load is unconditional (clamped address) and the factor selects --
    // under a branch hipcc waits for every outstanding load of the batch before it (s_waitcnt vmcnt(0)): one load in flight
    const int Ho = H >> 1, Wo = W >> 1, to = t >> 1, fo = f >> 1;
    const float k = (to < Ho && fo < Wo) ? 0.25f : 0.f;
    const int tc = to < Ho ? to : Ho - 1, fc = fo < Wo ? fo : Wo - 1;
    float d[8];
    ld8<T>(da + (((size_t)b * Ho + tc) * Wo + fc) * C + cg * 8, d);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = k * d[j];
  } else {
    const int Ho = H >> 1, to = t >> 1;
    if (to >= Ho) {
#pragma unroll
      for (int j = 0; j < 8; ++j) g[j] = 0.f;
      return;
    }
    const size_t pix = ((size_t)b * Ho + to) * W + f;
    float d[8], ds[8];
    ld8<T>(da + pix * C + cg * 8, d);
    drop_scale8(dc, pix * C + cg * 8, ds);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = 0.5f * ds[j] * d[j];
  }
}

template <typename T, int SRC>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ z, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ demb, const T* __restrict__ da,
                                                            float* __restrict__ partial, int B, int H, int W, int C,
                                                            DropCfg dc, int pix_per_block) {
  extern __shared__ float red[];  // [PL][C][2]
  const int CG = C >> 3, PL = 256 / CG;
  const int tid = threadIdx.x, cg = tid % CG, pl = tid / CG;
  const size_t npix = (size_t)B * H * W;
  const size_t p0 = (size_t)blockIdx.x * pix_per_block;
  const size_t p1 = (p0 + pix_per_block < npix) ? p0 + pix_per_block : npix;
  float mu[8], is[8], gm[8], bt[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    mu[j] = mean[c]; is[j] = invstd[c]; gm[j] = gamma[c]; bt[j] = beta[c];
    s1[j] = 0.f; s2[j] = 0.f;
  }
  const float inv_h = 1.0f / (float)H;
  // Pixel index arithmetic in 32 bits (npix < 2^31, launcher-checked: the 64-bit divisions by W and H cost more than the rest of
  // the loop body) and four pixels per trip, all eight loads requested before the first is used (one dependent pair at a time
  // held this pass at ~1 TB/s)
  auto body = [&](const float (&v)[8], const float (&g)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (v[j] - mu[j]) * is[j];
      const float dy = (fmaf(gm[j], xh, bt[j]) > 0.f) ? g[j] : 0.f;
      s1[j] += dy;
      s2[j] = fmaf(dy, xh, s2[j]);
    }
  };
  auto fetch = [&](unsigned p, float (&v)[8], float (&g)[8]) {
    const unsigned f = p % (unsigned)W, bt_ = p / (unsigned)W;
    const unsigned t = bt_ % (unsigned)H, b = bt_ / (unsigned)H;
    ld8<T>(z + (size_t)p * C + cg * 8, v);
    upstream8<T, SRC>(demb, da, dc, (int)b, (int)t, (int)f, cg, H, W, C, inv_h, g);
  };
  unsigned p = (unsigned)p0 + pl;
  const unsigned pe = (unsigned)p1;
  for (; p + 3u * PL < pe; p += 4u * PL) {
    float v[4][8], g[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) fetch(p + u * PL, v[u], g[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) body(v[u], g[u]);
  }
  for (; p < pe; p += PL) {
    float v[8], g[8];
    fetch(p, v, g);
    body(v, g);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[(pl * C + cg * 8 + j) * 2] = s1[j]; red[(pl * C + cg * 8 + j) * 2 + 1] = s2[j]; }
  __syncthreads();
  for (int e = tid; e < C * 2; e += 256) {
    float s = 0.f;
    for (int q = 0; q < PL; ++q) s += red[q * C * 2 + e];
    partial[(size_t)blockIdx.x * C * 2 + e] = s;
  }
}

template <typename T, int SRC>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ sums,
                                                           const float* __restrict__ demb, const T* __restrict__ da,
                                                           T* __restrict__ dz, int B, int H, int W, int C, DropCfg dc,
                                                           float inv_n, int pix_per_block) {
  // thread = (channel octet cg, pixel lane pl): the per-channel coefficients are loaded once and reused over the block's
  // pixels (one thread per 16-byte chunk spent 48 parameter loads on every data load)
  const int CG = C >> 3, PL = 256 / CG;
  const int tid = threadIdx.x, cg = tid % CG, pl = tid / CG;
  const size_t npix = (size_t)B * H * W;
  const size_t p0 = (size_t)blockIdx.x * pix_per_block;
  const size_t p1 = (p0 + pix_per_block < npix) ? p0 + pix_per_block : npix;
  float mu[8], is[8], gm[8], bt[8], k0[8], k1[8], k2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    mu[j] = mean[c]; is[j] = invstd[c]; gm[j] = gamma[c]; bt[j] = beta[c];
    k0[j] = gm[j] * is[j];
    k1[j] = sums[2 * c] * inv_n;
    k2[j] = sums[2 * c + 1] * inv_n;
  }
  const float inv_h = 1.0f / (float)H;
#pragma unroll 4
  for (unsigned p = (unsigned)p0 + pl; p < (unsigned)p1; p += PL) {          // (32-bit pixel arithmetic, see the reduce kernel)
    const int f = (int)(p % (unsigned)W);
    const unsigned bt_ = p / (unsigned)W;
    const int t = (int)(bt_ % (unsigned)H), b = (int)(bt_ / (unsigned)H);
    float v[8], g[8], o[8];
    ld8<T>(z + (size_t)p * C + cg * 8, v);
    upstream8<T, SRC>(demb, da, dc, b, t, f, cg, H, W, C, inv_h, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (v[j] - mu[j]) * is[j];
      const float dy = (fmaf(gm[j], xh, bt[j]) > 0.f) ? g[j] : 0.f;
      o[j] = k0[j] * (dy - k1[j] - xh * k2[j]);
    }
    st8<T>(dz + (size_t)p * C + cg * 8, o);
  }
}

// ---- apply pass of a ConvTranspose2d(k2, s2) + BatchNorm layer (auto-encoder decoder, SRC_DIRECT) that also does what its two
// consumers used to do in passes of their own: dz is written straight in the PATCH-MAJOR order the layer's gradient GEMMs read
// (zp[(b, t/2, f/2)][q = 2 (t&1) + (f&1)][C], `pixel_unshuffle_kernel`; a trailing output_padding column has no input pixel and is
// not stored) and its per-channel sums -- the convolution's bias gradient, every output pixel included -- leave as one [C] record
// per workgroup (summed as stored: after the rounding to T).  Same per-element formulas as bn_bwd_apply_kernel<T, SRC_DIRECT>.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_unshuffle_kernel(const T* __restrict__ z, const float* __restrict__ mean,
                                                                     const float* __restrict__ invstd,
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta, const float* __restrict__ sums,
                                                                     const T* __restrict__ da, T* __restrict__ zp,
                                                                     float* __restrict__ bias_partial, int B, int H, int W, int C,
                                                                     int Wh, float inv_n, int pix_per_block) {
  extern __shared__ float red[];  // [PL][C]
  const int CG = C >> 3, PL = 256 / CG;
  const int tid = threadIdx.x, cg = tid % CG, pl = tid / CG;
  const size_t npix = (size_t)B * H * W;
  const size_t p0 = (size_t)blockIdx.x * pix_per_block;
  const size_t p1 = (p0 + pix_per_block < npix) ? p0 + pix_per_block : npix;
  float mu[8], is[8], gm[8], bt[8], k0[8], k1[8], k2[8], bs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    mu[j] = mean[c]; is[j] = invstd[c]; gm[j] = gamma[c]; bt[j] = beta[c];
    k0[j] = gm[j] * is[j];
    k1[j] = sums[2 * c] * inv_n;
    k2[j] = sums[2 * c + 1] * inv_n;
    bs[j] = 0.f;
  }
  const unsigned Hh = (unsigned)H >> 1;
#pragma unroll 4
  for (unsigned p = (unsigned)p0 + pl; p < (unsigned)p1; p += PL) {
    const unsigned f = p % (unsigned)W, bt_ = p / (unsigned)W;
    const unsigned t = bt_ % (unsigned)H, b = bt_ / (unsigned)H;
    float v[8], g[8], o[8];
    ld8<T>(z + (size_t)p * C + cg * 8, v);
    ld8<T>(da + (size_t)p * C + cg * 8, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (v[j] - mu[j]) * is[j];
      const float dy = (fmaf(gm[j], xh, bt[j]) > 0.f) ? g[j] : 0.f;
      o[j] = k0[j] * (dy - k1[j] - xh * k2[j]);
      bs[j] += (sizeof(T) == 2) ? bf16_to_float(float_to_bf16(o[j])) : o[j];
    }
    if ((f >> 1) < (unsigned)Wh) {
      const size_t dst = ((((size_t)b * Hh + (t >> 1)) * Wh + (f >> 1)) * 4 + ((t & 1u) << 1) + (f & 1u)) * C + cg * 8;
      st8<T>(zp + dst, o);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[pl * C + cg * 8 + j] = bs[j];
  __syncthreads();
  for (int e = tid; e < C; e += 256) {
    float s = 0.f;
    for (int q = 0; q < PL; ++q) s += red[q * C + e];
    bias_partial[(size_t)blockIdx.x * C + e] = s;
  }
}

// ---- SRC_POOL reduction and apply pass on ROW PAIRS (H even): a thread takes a pooled pixel (to, f) x 8 channels -- ONE upstream load and ONE dropout
// draw serve the two rows 2*to, 2*to+1 of z that were averaged into it; several pooled pixels in flight per thread.  Same
// per-element formulas as the generic kernel (which ran one dependent 16-byte load at a time: 3.0 TB/s).
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_pool_kernel(const T* __restrict__ z, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const T* __restrict__ da,
                                                                 float* __restrict__ partial, int B, int H, int W, int C,
                                                                 DropCfg dc, int pp_per_block) {
  extern __shared__ float red[];  // [PL][C][2]
  const int CG = C >> 3, PL = 256 / CG, Ho = H >> 1;
  const int tid = threadIdx.x, cg = tid % CG, pl = tid / CG;
  const size_t npp = (size_t)B * Ho * W;
  const size_t p0 = (size_t)blockIdx.x * pp_per_block;
  const size_t p1 = (p0 + pp_per_block < npp) ? p0 + pp_per_block : npp;
  float mu[8], is[8], gm[8], bt[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    mu[j] = mean[c]; is[j] = invstd[c]; gm[j] = gamma[c]; bt[j] = beta[c];
    s1[j] = 0.f; s2[j] = 0.f;
  }
#pragma unroll 2
  for (size_t pp = p0 + pl; pp < p1; pp += PL) {
    const int f = (int)(pp % W);
    const size_t bt_ = pp / W;
    const int to = (int)(bt_ % Ho), b = (int)(bt_ / Ho);
    const size_t zp = ((size_t)b * H + 2 * to) * W + f;
    float v0[8], v1[8], d[8], ds[8];
    ld8<T>(z + zp * C + cg * 8, v0);
    ld8<T>(z + (zp + W) * C + cg * 8, v1);
    ld8<T>(da + pp * C + cg * 8, d);
    drop_scale8(dc, pp * C + cg * 8, ds);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float g = 0.5f * ds[j] * d[j];
      const float xh0 = (v0[j] - mu[j]) * is[j], xh1 = (v1[j] - mu[j]) * is[j];
      const float dy0 = (fmaf(gm[j], xh0, bt[j]) > 0.f) ? g : 0.f;
      const float dy1 = (fmaf(gm[j], xh1, bt[j]) > 0.f) ? g : 0.f;
      s1[j] += dy0;
      s2[j] = fmaf(dy0, xh0, s2[j]);
      s1[j] += dy1;
      s2[j] = fmaf(dy1, xh1, s2[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[(pl * C + cg * 8 + j) * 2] = s1[j]; red[(pl * C + cg * 8 + j) * 2 + 1] = s2[j]; }
  __syncthreads();
  for (int e = tid; e < C * 2; e += 256) {
    float s = 0.f;
    for (int q = 0; q < PL; ++q) s += red[q * C * 2 + e];
    partial[(size_t)blockIdx.x * C * 2 + e] = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_pool_kernel(const T* __restrict__ z, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ sums,
                                                                const T* __restrict__ da, T* __restrict__ dz, int B, int H,
                                                                int W, int C, DropCfg dc, float inv_n, int pp_per_block) {
  const int CG = C >> 3, PL = 256 / CG, Ho = H >> 1;
  const int tid = threadIdx.x, cg = tid % CG, pl = tid / CG;
  const size_t npp = (size_t)B * Ho * W;
  const size_t p0 = (size_t)blockIdx.x * pp_per_block;
  const size_t p1 = (p0 + pp_per_block < npp) ? p0 + pp_per_block : npp;
  float mu[8], is[8], gm[8], bt[8], k0[8], k1[8], k2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    mu[j] = mean[c]; is[j] = invstd[c]; gm[j] = gamma[c]; bt[j] = beta[c];
    k0[j] = gm[j] * is[j];
    k1[j] = sums[2 * c] * inv_n;
    k2[j] = sums[2 * c + 1] * inv_n;
  }
#pragma unroll 2
  for (size_t pp = p0 + pl; pp < p1; pp += PL) {
    const int f = (int)(pp % W);
    const size_t bt_ = pp / W;
    const int to = (int)(bt_ % Ho), b = (int)(bt_ / Ho);
    const size_t zp = ((size_t)b * H + 2 * to) * W + f;
    float v0[8], v1[8], d[8], ds[8], o0[8], o1[8];
    ld8<T>(z + zp * C + cg * 8, v0);
    ld8<T>(z + (zp + W) * C + cg * 8, v1);
    ld8<T>(da + pp * C + cg * 8, d);
    drop_scale8(dc, pp * C + cg * 8, ds);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float g = 0.5f * ds[j] * d[j];
      const float xh0 = (v0[j] - mu[j]) * is[j], xh1 = (v1[j] - mu[j]) * is[j];
      const float dy0 = (fmaf(gm[j], xh0, bt[j]) > 0.f) ? g : 0.f;
      const float dy1 = (fmaf(gm[j], xh1, bt[j]) > 0.f) ? g : 0.f;
      o0[j] = k0[j] * (dy0 - k1[j] - xh0 * k2[j]);
      o1[j] = k0[j] * (dy1 - k1[j] - xh1 * k2[j]);
    }
    st8<T>(dz + zp * C + cg * 8, o0);
    st8<T>(dz + (zp + W) * C + cg * 8, o1);
  }
}

// ---- SRC_MEANT apply walking down T: a thread owns (b, f, 8 channels); the upstream gradient g = demb / H does not depend on
// t, so it is loaded once (the generic kernel re-read 32 bytes of demb beside every 16 bytes of z) and there is no index
// arithmetic in the loop; eight row loads in flight per thread.  Same per-element formulas as bn_bwd_apply_kernel.
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_meant_kernel(const T* __restrict__ z, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ sums,
                                                                 const float* __restrict__ demb, T* __restrict__ dz, int B,
                                                                 int H, int W, int C, float inv_n) {
  const int CG = C >> 3;
  const size_t total = (size_t)B * W * CG;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int cg = (int)(i % CG);
  const int f = (int)((i / CG) % W), b = (int)(i / ((size_t)CG * W));
  float mu[8], is[8], gm[8], bt[8], k0[8], k1[8], k2[8], g[8];
  ld8<float>(demb + ((size_t)b * W + f) * C + cg * 8, g);
  const float inv_h = 1.0f / (float)H;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cg * 8 + j;
    mu[j] = mean[c]; is[j] = invstd[c]; gm[j] = gamma[c]; bt[j] = beta[c];
    k0[j] = gm[j] * is[j];
    k1[j] = sums[2 * c] * inv_n;
    k2[j] = sums[2 * c + 1] * inv_n;
    g[j] = g[j] * inv_h;
  }
  const size_t base = (((size_t)b * H) * W + f) * C + cg * 8, stride = (size_t)W * C;
#pragma unroll 8
  for (int t = 0; t < H; ++t) {
    float v[8], o[8];
    ld8<T>(z + base + t * stride, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = (v[j] - mu[j]) * is[j];
      const float dy = (fmaf(gm[j], xh, bt[j]) > 0.f) ? g[j] : 0.f;
      o[j] = k0[j] * (dy - k1[j] - xh * k2[j]);
    }
    st8<T>(dz + base + t * stride, o);
  }
}

// ---- loss: BCEWithLogitsLoss(mean) on smoothed labels (src/train.py:311-320) + its gradient
__global__ void bce_smooth_kernel(const float* __restrict__ logits, const float* __restrict__ labels, float eps, int B,
                                  float* __restrict__ loss, float* __restrict__ dlogits) {
  __shared__ double red[256];
  const int tid = threadIdx.x;
  double s = 0.0;
  for (int b = tid; b < B; b += 256) {
    const float z = logits[b];
    float y = labels[b];
    if (eps > 0.f) y = y * (1.0f - eps) + 0.5f * eps;
    s += (double)(fmaxf(z, 0.f) - z * y + log1pf(expf(-fabsf(z))));
    if (dlogits) dlogits[b] = (1.0f / (1.0f + expf(-z)) - y) / (float)B;
  }
  red[tid] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) red[tid] += red[tid + off];
    __syncthreads();
  }
  if (tid == 0 && loss) loss[0] = (float)(red[0] / (double)B);
}

// ---- optimiser: torch.optim.AdamW step on one flat buffer (decoupled weight decay, bias correction)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                             float bc1, float sqrt_bc2, float grad_scale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * grad_scale;
  float pi = p[i] * (1.0f - lr * wd);
  const float mi = b1 * m[i] + (1.0f - b1) * gi;
  const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
  const float denom = sqrtf(vi) / sqrt_bc2 + eps;
  pi -= (lr / bc1) * (mi / denom);
  p[i] = pi; m[i] = mi; v[i] = vi;
}

// ================================================================================================ launchers
hipError_t launch_bn_finalize(const float* partial, int nparts, int C, double n, float* mean, float* var, float* invstd,
                              float* running_mean, float* running_var, float momentum, hipStream_t s) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, s, partial, nparts, C, n, mean, var, invstd, running_mean,
                     running_var, momentum);
  return hipGetLastError();
}

// scratch: >= 64 * n floats, only needed when nparts > 2048 (second reduction level)
hipError_t launch_reduce_partials(const float* partial, int nparts, int n, float scale, float* out, hipStream_t s,
                                  float* scratch) {
  const int gx = (n + 31) / 32;
  if (nparts <= 256 || !scratch) {      // (a single level walks nparts / 8 records per thread: 83 us for 1800 records of 128 floats)
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, 1), dim3(256), 0, s, partial, nparts, n, scale, out, nparts);
    return hipGetLastError();
  }
  const int nchunks = 64, chunk = (nparts + nchunks - 1) / nchunks;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, nchunks), dim3(256), 0, s, partial, nparts, n, 1.0f, scratch, chunk);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(gx, 1), dim3(256), 0, s, scratch, nchunks, n, scale, out, nchunks);
  return hipGetLastError();
}

hipError_t launch_reduce_partials_strided(const float* partial, int nparts, int stride, int off, int n, float* out,
                                          hipStream_t s) {
  hipLaunchKernelGGL(reduce_partials_strided_kernel, dim3((n + 255) / 256), dim3(256), 0, s, partial, nparts, stride, off,
                     n, out);
  return hipGetLastError();
}

hipError_t launch_reduce_wgrad_window(const float* partial, int nparts, int stride, int cin, int cout, int cin_total,
                                      int ci_off, int co_off, float* dw, hipStream_t s) {
  const int n = cout * cin * 9;
  hipLaunchKernelGGL(reduce_wgrad_window_kernel, dim3((n + 255) / 256), dim3(256), 0, s, partial, nparts, stride, cin, cout,
                     cin_total, ci_off, co_off, dw);
  return hipGetLastError();
}

hipError_t launch_reduce_wgrad_record(const float* partial, int nparts, int stride, int cin, int cout, int cin_total,
                                      int ci_off, int co_off, float* dw, float* db, hipStream_t s, int perm) {
  const int total = cout * cin * 9 + cout;
  hipLaunchKernelGGL(reduce_wgrad_record_kernel, dim3((total + 63) / 64), dim3(256), 0, s, partial, nparts, stride, cin, cout,
                     cin_total, ci_off, co_off, dw, db, perm);
  return hipGetLastError();
}

hipError_t launch_bn_relu_pool_drop(int prec, const void* z, const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, void* out, int B, int H, int W, int C, const DropCfg& dc,
                                    hipStream_t s) {
  const size_t total = (size_t)B * (H / 2) * W * (C / 8);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (prec == DFA_PREC_BF16)
    hipLaunchKernelGGL(bn_relu_poolh2_drop_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)z, mean, invstd, gamma, beta, (bf16_t*)out, B, H, W, C, dc);
  else
    hipLaunchKernelGGL(bn_relu_poolh2_drop_kernel<float>, grid, block, 0, s, (const float*)z, mean, invstd, gamma, beta, (float*)out, B, H, W, C, dc);
  return hipGetLastError();
}

int cl_stats_blocks(size_t npix, int* pix_per_block) {
  *pix_per_block = 1024;
  return (int)((npix + 1023) / 1024);
}

hipError_t launch_cl_stats(int prec, const void* z, float* partial, size_t npix, int C, hipStream_t s) {
  int ppb;
  const int nblk = cl_stats_blocks(npix, &ppb);
  const int PL = 256 / (C / 8);
  const size_t lds = (size_t)PL * C * 2 * sizeof(float);
  if (prec == DFA_PREC_BF16)
    hipLaunchKernelGGL(cl_stats_kernel<bf16_t>, dim3(nblk), dim3(256), lds, s, (const bf16_t*)z, partial, npix, C, ppb);
  else
    hipLaunchKernelGGL(cl_stats_kernel<float>, dim3(nblk), dim3(256), lds, s, (const float*)z, partial, npix, C, ppb);
  return hipGetLastError();
}

hipError_t launch_bn_relu_pool(int prec, int pool, const void* z, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, void* out, int B, int H, int W, int C,
                               hipStream_t s) {
  const size_t total = (size_t)B * (H / pool) * (W / pool) * (C / 8);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (prec == DFA_PREC_BF16) {
    if (pool == 2) hipLaunchKernelGGL((bn_relu_pool_kernel<bf16_t, 2>), grid, block, 0, s, (const bf16_t*)z, mean, invstd, gamma, beta, (bf16_t*)out, B, H, W, C);
    else hipLaunchKernelGGL((bn_relu_pool_kernel<bf16_t, 1>), grid, block, 0, s, (const bf16_t*)z, mean, invstd, gamma, beta, (bf16_t*)out, B, H, W, C);
  } else {
    if (pool == 2) hipLaunchKernelGGL((bn_relu_pool_kernel<float, 2>), grid, block, 0, s, (const float*)z, mean, invstd, gamma, beta, (float*)out, B, H, W, C);
    else hipLaunchKernelGGL((bn_relu_pool_kernel<float, 1>), grid, block, 0, s, (const float*)z, mean, invstd, gamma, beta, (float*)out, B, H, W, C);
  }
  return hipGetLastError();
}

hipError_t launch_bn_relu_meant(int prec, const void* z, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, float* emb, int B, int H, int W, int C, hipStream_t s, float* msum) {
  const size_t total = (size_t)B * W * (C / 8);
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define DFA_BRM(TT, MS) hipLaunchKernelGGL((bn_relu_meant_kernel<TT, MS>), grid, block, 0, s, (const TT*)z, mean, invstd, gamma, beta, emb, msum, B, H, W, C)
  if (prec == DFA_PREC_BF16) { if (msum) DFA_BRM(bf16_t, true); else DFA_BRM(bf16_t, false); }
  else { if (msum) DFA_BRM(float, true); else DFA_BRM(float, false); }
#undef DFA_BRM
  return hipGetLastError();
}

// BatchNorm backward of the mean-over-T layer with the reduction taken from the forward's saved sums (msum[B][W][C][2]):
// reduce (3 floats per position) -> fixed-order second stage -> the usual apply pass (z -> dz).
hipError_t launch_bn_bwd_meant_saved(int prec, const void* z, const float* mean, const float* invstd, const float* gamma,
                                     const float* beta, const float* demb, const float* msum, float* partial, float* sums,
                                     void* dz, int B, int H, int W, int C, hipStream_t s, const BnSync* sync) {
  const size_t npos = (size_t)B * W;
  const int PL = 256 / (C / 8);
  const int ppb = 8 * PL;
  const int nblk = (int)((npos + ppb - 1) / ppb);
  const size_t lds = (size_t)PL * C * 2 * sizeof(float);
  hipLaunchKernelGGL(bn_bwd_reduce_saved_kernel, dim3(nblk), dim3(256), lds, s, demb, msum, partial, npos, C, 1.0f / (float)H, ppb);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  e = launch_reduce_partials(partial, nblk, C * 2, 1.0f, sums, s, nullptr);
  if (e != hipSuccess) return e;
  const float* sums_a;
  float isc;
  e = bn_sync_sums(sync, sums, C * 2, s, &sums_a, &isc);          // synchronised BatchNorm: global sums, global count
  if (e != hipSuccess) return e;
  dim3 g2((unsigned)(((size_t)B * W * (C / 8) + 255) / 256));
  const float inv_n = (float)(1.0 / ((double)B * H * W)) * isc;
  if (prec == DFA_PREC_BF16)
    hipLaunchKernelGGL(bn_bwd_apply_meant_kernel<bf16_t>, g2, dim3(256), 0, s, (const bf16_t*)z, mean, invstd, gamma, beta, sums_a,
                       demb, (bf16_t*)dz, B, H, W, C, inv_n);
  else
    hipLaunchKernelGGL(bn_bwd_apply_meant_kernel<float>, g2, dim3(256), 0, s, (const float*)z, mean, invstd, gamma, beta, sums_a,
                       demb, (float*)dz, B, H, W, C, inv_n);
  return hipGetLastError();
}

hipError_t launch_linear_bwd(const float* dlogits, const float* w, const float* emb, float* demb, float* dw, float* db,
                             int B, int K, hipStream_t s, int tc, int tw) {
  hipLaunchKernelGGL(linear_bwd_kernel, dim3((K + 31) / 32), dim3(256), 0, s, dlogits, w, emb, demb, dw, db, B, K, tc, tw);
  return hipGetLastError();
}

hipError_t bn_sync_sums(const BnSync* sy, const float* sums, int count, hipStream_t s, const float** sums_apply, float* inv_scale) {
  *sums_apply = sums;
  *inv_scale = 1.0f;
  if (!sy || !sy->fn) return hipSuccess;
  hipError_t e = hipMemcpyAsync(sy->buf, sums, (size_t)count * sizeof(float), hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return e;
  if (sy->fn(sy->user, sy->buf, count) != 0) return hipErrorUnknown;     // the caller's collective failed
  *sums_apply = sy->buf;
  *inv_scale = 1.0f / (float)sy->world;
  return hipSuccess;
}

int bn_bwd_blocks(int B, int H, int W, int* pix_per_block) {
  const size_t npix = (size_t)B * H * W;
  int ppb = 4096;
  *pix_per_block = ppb;
  return (int)((npix + ppb - 1) / ppb);
}

// sums: [C][2] (S1 = dbeta, S2 = dgamma) written by the reduce stage; dz may alias nothing.  scratch: 64 * C * 2 floats for the
// two-level reduction of the block records (nullptr: one level)
hipError_t launch_bn_bwd(int prec, int src, const void* z, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, const float* demb, const void* da, float* partial, float* sums, void* dz,
                         int B, int H, int W, int C, const DropCfg& dc, hipStream_t s, float* scratch, const BnSync* sync,
                         BnBwdFold* fold) {
  int ppb;
  int nblk = bn_bwd_blocks(B, H, W, &ppb);
  if (fold && (src != SRC_DIRECT || (H & 1) || 2 * fold->Wh > W || !fold->zp || !fold->bias_partial)) return hipErrorInvalidValue;
  if ((size_t)B * H * W >= ((size_t)1 << 31)) return hipErrorInvalidValue;     // 32-bit pixel indices in the generic kernels
  if (src == SRC_DIRECT || src == SRC_POOL22) {   // the auto-encoder's layers (55-900 blocks of 4096 pixels did not fill the chip)
    ppb = 1024;
    nblk = (int)(((size_t)B * H * W + ppb - 1) / ppb);
  }
  const int PL = 256 / (C / 8);
  const size_t lds = (size_t)PL * C * 2 * sizeof(float);
  const int ppb2 = 16 * PL;                       // pixels per block of the apply pass: 16 chunks per thread
  dim3 g2((unsigned)(((size_t)B * H * W + ppb2 - 1) / ppb2));
  const float inv_n = (float)(1.0 / ((double)B * H * W));
  const float* sums_a = sums;      // what the apply stage reads: the all-reduced copy under synchronised BatchNorm
  float isc = 1.0f;
#define DFA_BN_BWD_POOL(TT)                                                                                            \
  do {                                                                                                                 \
    const int ppp = ppb / 2;                                                                                           \
    const int nb = (int)(((size_t)B * (H / 2) * W + ppp - 1) / ppp);   /* <= nblk: partial has room */                  \
    hipLaunchKernelGGL(bn_bwd_reduce_pool_kernel<TT>, dim3(nb), dim3(256), lds, s, (const TT*)z, mean, invstd, gamma, beta, \
                       (const TT*)da, partial, B, H, W, C, dc, ppp);                                                   \
    hipError_t e = hipGetLastError();                                                                                  \
    if (e != hipSuccess) return e;                                                                                     \
    e = launch_reduce_partials(partial, nb, C * 2, 1.0f, sums, s, scratch);                                            \
    if (e != hipSuccess) return e;                                                                                     \
    e = bn_sync_sums(sync, sums, C * 2, s, &sums_a, &isc);                                                             \
    if (e != hipSuccess) return e;                                                                                     \
    if (dz) {   /* 0.435 vs 0.475 ms for the generic kernel at [256,160,180,64] */                                      \
      const int ppp2 = 8 * PL;                                                                                         \
      dim3 g3((unsigned)(((size_t)B * (H / 2) * W + ppp2 - 1) / ppp2));                                                \
      hipLaunchKernelGGL(bn_bwd_apply_pool_kernel<TT>, g3, dim3(256), 0, s, (const TT*)z, mean, invstd, gamma, beta, sums_a, \
                         (const TT*)da, (TT*)dz, B, H, W, C, dc, inv_n * isc, ppp2);                                   \
    }                                                                                                                  \
  } while (0)
#define DFA_BN_BWD(TT, SRC)                                                                                            \
  do {                                                                                                                 \
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<TT, SRC>), dim3(nblk), dim3(256), lds, s, (const TT*)z, mean, invstd, gamma, \
                       beta, demb, (const TT*)da, partial, B, H, W, C, dc, ppb);                                       \
    hipError_t e = hipGetLastError();                                                                                  \
    if (e != hipSuccess) return e;                                                                                     \
    e = launch_reduce_partials(partial, nblk, C * 2, 1.0f, sums, s, scratch);                                                     \
    if (e != hipSuccess) return e;                                                                                     \
    e = bn_sync_sums(sync, sums, C * 2, s, &sums_a, &isc);                                                             \
    if (e != hipSuccess) return e;                                                                                     \
    if (fold) {                                                                                                        \
      fold->nrec = (int)g2.x;                                                                                          \
      hipLaunchKernelGGL((bn_bwd_apply_unshuffle_kernel<TT>), g2, dim3(256), lds / 2, s, (const TT*)z, mean, invstd, gamma, beta, \
                         sums_a, (const TT*)da, (TT*)fold->zp, fold->bias_partial, B, H, W, C, fold->Wh, inv_n * isc, ppb2);  \
    } else if (dz)                                                                                                     \
      hipLaunchKernelGGL((bn_bwd_apply_kernel<TT, SRC>), g2, dim3(256), 0, s, (const TT*)z, mean, invstd, gamma, beta, sums_a, \
                         demb, (const TT*)da, (TT*)dz, B, H, W, C, dc, inv_n * isc, ppb2);                                  \
  } while (0)
  if (prec == DFA_PREC_BF16) {
    if (src == SRC_MEANT) DFA_BN_BWD(bf16_t, SRC_MEANT);
    else if (src == SRC_POOL && !(H & 1)) DFA_BN_BWD_POOL(bf16_t);
    else if (src == SRC_POOL) DFA_BN_BWD(bf16_t, SRC_POOL);
    else if (src == SRC_DIRECT) DFA_BN_BWD(bf16_t, SRC_DIRECT);
    else DFA_BN_BWD(bf16_t, SRC_POOL22);
  } else {
    if (src == SRC_MEANT) DFA_BN_BWD(float, SRC_MEANT);
    else if (src == SRC_POOL && !(H & 1)) DFA_BN_BWD_POOL(float);
    else if (src == SRC_POOL) DFA_BN_BWD(float, SRC_POOL);
    else if (src == SRC_DIRECT) DFA_BN_BWD(float, SRC_DIRECT);
    else DFA_BN_BWD(float, SRC_POOL22);
  }
#undef DFA_BN_BWD
#undef DFA_BN_BWD_POOL
  return hipGetLastError();
}

hipError_t launch_bce_smooth(const float* logits, const float* labels, float eps, int B, float* loss, float* dlogits,
                             hipStream_t s) {
  hipLaunchKernelGGL(bce_smooth_kernel, dim3(1), dim3(256), 0, s, logits, labels, eps, B, loss, dlogits);
  return hipGetLastError();
}

hipError_t launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                        float wd, int step, float grad_scale, hipStream_t s) {
  const float bc1 = (float)(1.0 - pow((double)b1, (double)step));
  const float sqrt_bc2 = (float)sqrt(1.0 - pow((double)b2, (double)step));
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, g, m, v, n, lr, b1, b2, eps, wd,
                     bc1, sqrt_bc2, grad_scale);
  return hipGetLastError();
}

}  // namespace dfa
