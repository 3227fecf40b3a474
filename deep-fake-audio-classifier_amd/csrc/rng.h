// rng.h -- counter-based Philox4x32 for the dropout masks (src/model.py:19,25 nn.Dropout in train mode).
// The mask of element `idx` of dropout layer `layer` is a pure function of (seed, offset, layer, idx), so the backward
// pass regenerates it instead of storing it.  torch's CPU generator stream cannot be reproduced on a GPU; parity for
// training is exact only with dropout = 0 and statistical otherwise (SURVEY.md section 7 "hard parts").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dfa {

// R rounds.  The dropout masks use 7 -- the fewest rounds at which Philox4x32 passes BigCrush (Salmon et al., SC'11, table 2);
// the standard 10 are a safety margin the masks do not need, and the rounds are quarter-rate integer multiplies that bound the
// data-gradient kernel carrying block 1's mask in its epilogue (one call per 16-byte store).  The augmentation noise keeps 10.
__device__ __forceinline__ void philox_round(uint4& c, uint2& k) {
  // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a v_mul_hi_u32 / v_mul_lo_u32 pair: both are
  // quarter-rate instructions, and the multiplies are what a mask costs
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x, p1 = (unsigned long long)0xCD9E8D57u * c.z;
  const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
  c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
  k.x += 0x9E3779B9u;
  k.y += 0xBB67AE85u;
}
template <int R>
__device__ __forceinline__ uint4 philox4x32(uint4 c, uint2 k) {
#pragma unroll
  for (int r = 0; r < R; ++r) philox_round(c, k);
  return c;
}

constexpr int kDropRounds = 7;

struct DropCfg {
  unsigned thresh;   // drop when r < thresh (thresh = p * 2^32); 0 disables dropout
  float scale;       // 1 / (1 - p)
  uint64_t seed, offset;
  unsigned layer;
};

// keep-scale factors (0 or scale) of 8 consecutive elements starting at idx (idx % 8 == 0).  ONE Philox call per 8
// elements: each 32-bit output word gives two 16-bit uniforms, compared with the top 16 bits of the threshold (the drop
// probability is quantised to 1/65536 -- the mask stream of the reference's generator cannot be reproduced anyway, parity
// with dropout > 0 is statistical).  The Philox rounds are quarter-rate integer multiplies: with two calls per 16-byte
// chunk the elementwise dropout passes were RNG-bound at ~1.2 TB/s.
__device__ __forceinline__ void drop_scale8(const DropCfg& d, uint64_t idx, float* f) {
  if (d.thresh == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = 1.f;
    return;
  }
  const uint64_t q = (idx >> 3) + d.offset;
  const uint2 key = make_uint2((unsigned)d.seed, (unsigned)(d.seed >> 32));
  const uint4 r0 = philox4x32<kDropRounds>(make_uint4((unsigned)q, (unsigned)(q >> 32), d.layer, 0u), key);
  const unsigned t16 = d.thresh >> 16;
  const unsigned r[4] = {r0.x, r0.y, r0.z, r0.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f[2 * j] = ((r[j] & 0xffffu) < t16) ? 0.f : d.scale;
    f[2 * j + 1] = ((r[j] >> 16) < t16) ? 0.f : d.scale;
  }
}

// the same draw as drop_scale8 as AND-masks over packed bf16 pairs: km[j] covers elements 2j (low half) and 2j+1 (high half),
// all ones where the element is kept
// (the same draw in pieces, for callers that spread the rounds between other work: counter / key, kDropRounds x
//  philox_round, then drop_keep_from)
__device__ __forceinline__ void drop_counter(const DropCfg& d, uint64_t idx, uint4& c, uint2& k) {
  const uint64_t q = (idx >> 3) + d.offset;
  k = make_uint2((unsigned)d.seed, (unsigned)(d.seed >> 32));
  c = make_uint4((unsigned)q, (unsigned)(q >> 32), d.layer, 0u);
}
__device__ __forceinline__ void drop_keep_from(const DropCfg& d, const uint4& r0, unsigned* km) {
  const unsigned t16 = d.thresh >> 16;
  const unsigned r[4] = {r0.x, r0.y, r0.z, r0.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) km[j] = (((r[j] & 0xffffu) < t16) ? 0u : 0x0000ffffu) | (((r[j] >> 16) < t16) ? 0u : 0xffff0000u);
}
__device__ __forceinline__ void drop_keep8(const DropCfg& d, uint64_t idx, unsigned* km) {
  const uint64_t q = (idx >> 3) + d.offset;
  const uint2 key = make_uint2((unsigned)d.seed, (unsigned)(d.seed >> 32));
  const uint4 r0 = philox4x32<kDropRounds>(make_uint4((unsigned)q, (unsigned)(q >> 32), d.layer, 0u), key);
  const unsigned t16 = d.thresh >> 16;
  const unsigned r[4] = {r0.x, r0.y, r0.z, r0.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) km[j] = (((r[j] & 0xffffu) < t16) ? 0u : 0x0000ffffu) | (((r[j] >> 16) < t16) ? 0u : 0xffff0000u);
}

// ---- train-time feature augmentation folded into the loads of the kernels that read x (src/train.py:68-69 applies
// src/augmentation.py:5-186 to the batch before the model).  The augmented tensor is
//   xa[b][t][f] = keep[f] * mask(x[b][(t - shift) mod T][f]) + std * N(0,1)(seed, offset + (b*T + t)*F + f)
// (mask spans refer to frames / feature dims BEFORE the shift: the reference's op order time mask, feature mask, roll,
// channel drop, jitter).  on = 0: identity.
struct AugCfg {
  int on;
  int T, F;
  int shift;                                // normalised into [0, T)
  const float* keep;                        // [F] multiplicative mask or null
  int tm_start, tm_len, fm_start, fm_len;   // zeroed spans (len 0 = none)
  float std;
  uint64_t seed, offset;
};

__device__ __forceinline__ float aug_noise(const AugCfg& a, uint64_t idx) {
  const uint64_t q = idx + a.offset;
  const uint4 r = philox4x32<10>(make_uint4((unsigned)q, (unsigned)(q >> 32), 0x41554721u, 0u),
                                make_uint2((unsigned)a.seed, (unsigned)(a.seed >> 32)));
  const float u0 = ((float)(r.x >> 8) + 1.0f) * (1.0f / 16777216.0f), u1 = (float)(r.y >> 8) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * __logf(u0)) * __cosf(6.28318530717958648f * u1) * a.std;   // Box-Muller, one normal per element
}
// source frame of output frame t (torch.roll(shifts = shift, dims = 1)); t in [0, T)
__device__ __forceinline__ int aug_src_t(const AugCfg& a, int t) {
  if (!a.on) return t;
  int ts = t - a.shift;
  return ts < 0 ? ts + a.T : ts;
}
// finish an element: xraw = x[b][aug_src_t(t)][f] as float
__device__ __forceinline__ float aug_apply(const AugCfg& a, float xraw, int b, int t, int f) {
  if (!a.on) return xraw;
  const int ts = aug_src_t(a, t);
  const bool masked = (a.tm_len > 0 && ts >= a.tm_start && ts < a.tm_start + a.tm_len) ||
                      (a.fm_len > 0 && f >= a.fm_start && f < a.fm_start + a.fm_len);
  float v = masked ? 0.f : xraw;
  if (a.keep) v *= a.keep[f];
  if (a.std > 0.f) v += aug_noise(a, ((uint64_t)b * a.T + t) * a.F + f);
  return v;
}

}  // namespace dfa
