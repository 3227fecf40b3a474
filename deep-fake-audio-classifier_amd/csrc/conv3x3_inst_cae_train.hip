// conv3x3_inst_cae_train.hip -- conv3x3_mfma instantiations of the ConvAutoencoder training step: encoder blocks 2-4
// forward storing the pre-BatchNorm output, and their data gradients.  Wide layers are covered by 64-input-channel
// launches chained through fp32 partial sums (EPI_RAW / ACCIN), as in conv3x3_inst_cae.hip / conv3x3_inst_train.hip.
#include "dfa_internal.h"

namespace dfa {

// forward, pre-BN output z.  With a.stats_partial set (encoder blocks 2 and 3), the BatchNorm statistics ride on the epilogue as in
// the CNN2D's blocks 2 / 3 (records [B * strips][COUT][2], conv3x3_mfma.h; block 2 IS the CNN2D's block-2 kernel) -- no separate
// pass over z.  Block 4 (one 22-column strip per sample at F = 180: little work per workgroup) keeps the separate pass: the
// epilogue's fixed cost (+49 us) was twice the pass over its 118 MB output.
hipError_t launch_train_fwd2(int prec, const ConvArgs& a, hipStream_t s);
hipError_t launch_train_fwd3(int prec, const ConvArgs& a, hipStream_t s);
hipError_t launch_cae_train_fwd(int prec, int cin, const ConvArgs& a, float* raw_tmp, hipStream_t s, int wide) {
  const size_t es = (prec == DFA_PREC_BF16) ? 2 : 4;
  if (cin == 32) {
    if (a.stats_partial) return launch_train_fwd2(prec, a, s);
    if (prec == DFA_PREC_BF16) return launch_conv3x3<bf16_t, 32, 2, 2, 2, 1, EPI_PLAIN, 1>(a, s);
    return launch_conv3x3<float, 32, 2, 2, 2, 1, EPI_PLAIN, 1>(a, s);
  }
  if (cin == 64) {
    if (a.stats_partial) return launch_train_fwd3(prec, a, s);
    if (prec == DFA_PREC_BF16) return launch_conv3x3<bf16_t, 64, 4, 1, 1, 1, EPI_PLAIN, 1>(a, s);
    return launch_conv3x3<float, 64, 4, 1, 1, 1, EPI_PLAIN, 1, false, true>(a, s);
  }
  if (a.stats_partial) return hipErrorInvalidValue;
  // cin == 128, bf16, wide: ONE launch with the 288-register weight slice (one wave per SIMD, LDS-DMA staging: no spill), as the eval
  // forward's block 4; a.wpack is then ONE [COUT/32][9][8][64] image
  if (wide && prec == DFA_PREC_BF16) return launch_conv3x3<bf16_t, 128, 4, 1, 1, 1, EPI_PLAIN, 1, false, true>(a, s);
  // otherwise two 64-channel halves chained through fp32 partial sums; a.wpack holds the two images back to back
  const int nkg = (prec == DFA_PREC_BF16) ? 4 : 8;
  ConvArgs p1 = a, p2 = a;
  p1.in_pix_bytes = p2.in_pix_bytes = (int)(128 * es);
  p1.in_ch_off_bytes = 0;
  p2.in_ch_off_bytes = (int)(64 * es);
  p1.raw_out = raw_tmp;
  p2.acc_in = raw_tmp;
  p2.wpack = a.wpack + (size_t)(a.COUT / 32) * 9 * nkg * 64;
  hipError_t e;
  if (prec == DFA_PREC_BF16) {
    e = launch_conv3x3<bf16_t, 64, 4, 1, 1, 1, EPI_RAW, 1>(p1, s);
    if (e != hipSuccess) return e;
    return launch_conv3x3<bf16_t, 64, 4, 1, 1, 1, EPI_PLAIN, 1, true>(p2, s);
  }
  e = launch_conv3x3<float, 64, 4, 1, 1, 1, EPI_RAW, 1, false, true>(p1, s);
  if (e != hipSuccess) return e;
  return launch_conv3x3<float, 64, 4, 1, 1, 1, EPI_PLAIN, 1, true, true>(p2, s);
}

// data gradient of encoder block 4: dz4 [.,.,256] -> de3 [.,.,128]; four 64-channel launches chained through raw_tmp.
// a.wpack: four [128/32][9][64/KG][64] images (channel windows 0-63, 64-127, 128-191, 192-255 of the 256 inputs)
// wide (bf16): TWO 128-channel launches (a.wpack: two [128/32][9][8][64] images, channel windows 0-127, 128-255)
hipError_t launch_cae_dgrad4(int prec, const ConvArgs& a, float* raw_tmp, hipStream_t s, int wide) {
  const size_t es = (prec == DFA_PREC_BF16) ? 2 : 4;
  const int nkg = (prec == DFA_PREC_BF16) ? 4 : 8;
  if (wide && prec == DFA_PREC_BF16) {
    ConvArgs p1 = a, p2 = a;
    p1.in_pix_bytes = p2.in_pix_bytes = 256 * 2;
    p1.in_ch_off_bytes = 0;
    p2.in_ch_off_bytes = 128 * 2;
    p1.raw_out = raw_tmp;
    p2.acc_in = raw_tmp;
    p2.wpack = a.wpack + (size_t)(128 / 32) * 9 * 8 * 64;
    hipError_t e = launch_conv3x3<bf16_t, 128, 4, 1, 1, 1, EPI_RAW, 1, false, true>(p1, s);
    if (e != hipSuccess) return e;
    return launch_conv3x3<bf16_t, 128, 4, 1, 1, 1, EPI_PLAIN, 1, true, true>(p2, s);
  }
  const size_t img = (size_t)(128 / 32) * 9 * nkg * 64;
  for (int c = 0; c < 4; ++c) {
    ConvArgs p = a;
    p.in_pix_bytes = (int)(256 * es);
    p.in_ch_off_bytes = (int)(64 * c * es);
    p.wpack = a.wpack + img * c;
    p.raw_out = raw_tmp;
    p.acc_in = raw_tmp;
    hipError_t e;
    if (prec == DFA_PREC_BF16) {
      if (c == 0) e = launch_conv3x3<bf16_t, 64, 4, 1, 1, 1, EPI_RAW, 1>(p, s);
      else if (c < 3) e = launch_conv3x3<bf16_t, 64, 4, 1, 1, 1, EPI_RAW, 1, true>(p, s);
      else e = launch_conv3x3<bf16_t, 64, 4, 1, 1, 1, EPI_PLAIN, 1, true>(p, s);
    } else {
      if (c == 0) e = launch_conv3x3<float, 64, 4, 1, 1, 1, EPI_RAW, 1, false, true>(p, s);
      else if (c < 3) e = launch_conv3x3<float, 64, 4, 1, 1, 1, EPI_RAW, 1, true, true>(p, s);
      else e = launch_conv3x3<float, 64, 4, 1, 1, 1, EPI_PLAIN, 1, true, true>(p, s);
    }
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace dfa
