// conv3x3_inst_cae.hip -- MFMA instantiations for the ConvAutoencoder (src/model_cae.py:40-55, 63-76):
//   encoder blocks 2-4 = conv3x3_mfma with the AvgPool2d(2) epilogue; decoder blocks 1-3 = convt2x2_mfma.
// Encoder block 4 (Cin = 128): the bf16 weight slice (288 VGPRs) fits one launch at one wave per SIMD; in fp32 the
// 576-VGPR slice does not, so the launch is split over Cin: channels 0-63 -> raw fp32 partial sums, then channels
// 64-127 start from those sums and run the fused epilogue (bit-identical to one fp32 fma chain over all 1152 terms).
#include "dfa_internal.h"
#include "convt2x2_mfma.h"

namespace dfa {

// dma = 1 (bf16): the input ring is staged by LDS-DMA (global_load_lds), as the CNN2D blocks do; 0 = through registers
hipError_t launch_cae_enc2(int prec, const ConvArgs& a, hipStream_t s, int pipe, int dma) {
  if (prec == DFA_PREC_BF16 && dma && pipe) return launch_conv3x3<bf16_t, 32, 2, 2, 2, 1, EPI_POOL_2X2, 2, false, true>(a, s);
  if (prec == DFA_PREC_BF16)
    return pipe ? launch_conv3x3<bf16_t, 32, 2, 2, 2, 1, EPI_POOL_2X2, 2>(a, s)
                : launch_conv3x3<bf16_t, 32, 2, 2, 2, 1, EPI_POOL_2X2, 2, false, false, false, 0>(a, s);
  return launch_conv3x3<float, 32, 2, 2, 2, 1, EPI_POOL_2X2, 1>(a, s);
}

hipError_t launch_cae_enc3(int prec, const ConvArgs& a, hipStream_t s, int pipe, int dma) {
  if (prec == DFA_PREC_BF16 && dma && pipe) return launch_conv3x3<bf16_t, 64, 4, 1, 1, 1, EPI_POOL_2X2, 2, false, true>(a, s);
  if (prec == DFA_PREC_BF16)
    return pipe ? launch_conv3x3<bf16_t, 64, 4, 1, 1, 1, EPI_POOL_2X2, 2>(a, s)
                : launch_conv3x3<bf16_t, 64, 4, 1, 1, 1, EPI_POOL_2X2, 2, false, false, false, 0>(a, s);
  return launch_conv3x3<float, 64, 4, 1, 1, 1, EPI_POOL_2X2, 1>(a, s);
}

// a.wpack: bf16 -> one [256/32][9][8][64] image; fp32 -> two consecutive [256/32][9][8][64] images (Cin halves)
hipError_t launch_cae_enc4(int prec, const ConvArgs& a, float* raw_tmp, hipStream_t s, int dma) {
  if (prec == DFA_PREC_BF16 && dma) return launch_conv3x3<bf16_t, 128, 4, 1, 1, 1, EPI_POOL_2X2, 1, false, true>(a, s);
  if (prec == DFA_PREC_BF16) return launch_conv3x3<bf16_t, 128, 4, 1, 1, 1, EPI_POOL_2X2, 1>(a, s);
  ConvArgs p1 = a;
  p1.in_pix_bytes = 128 * 4;
  p1.in_ch_off_bytes = 0;
  p1.raw_out = raw_tmp;
  hipError_t e = launch_conv3x3<float, 64, 4, 1, 1, 1, EPI_RAW, 1>(p1, s);
  if (e != hipSuccess) return e;
  ConvArgs p2 = a;
  p2.in_pix_bytes = 128 * 4;
  p2.in_ch_off_bytes = 64 * 4;
  p2.acc_in = raw_tmp;
  p2.wpack = a.wpack + (size_t)(256 / 32) * 9 * 8 * 64;
  // train_conv_variant 7 / 8 (diagnostic, tools/gpu_accin_probe.py): the asm-pipelined forms (3 / 4 reads in flight) of this
  // one-wave-per-SIMD fp32 ACCIN kernel with 288 weight registers; never the default
  if (train_conv_variant() == 7) return launch_conv3x3<float, 64, 4, 1, 1, 1, EPI_POOL_2X2, 1, true, false, false, 3>(p2, s);
  if (train_conv_variant() == 8) return launch_conv3x3<float, 64, 4, 1, 1, 1, EPI_POOL_2X2, 1, true, false, false, 4>(p2, s);
  return launch_conv3x3<float, 64, 4, 1, 1, 1, EPI_POOL_2X2, 1, true>(p2, s);
}

hipError_t launch_cae_dec(int prec, int cin, const ConvTArgs& a, hipStream_t s) {
  if (a.stats_partial) {   // train mode: pre-BatchNorm output + per-workgroup statistics records (cae_dec_stats_records of them)
    if (prec == DFA_PREC_BF16) {
      if (cin == 256) return launch_convt2x2<bf16_t, 256, 4, true>(a, s);
      if (cin == 128) return launch_convt2x2<bf16_t, 128, 4, true>(a, s);
      return launch_convt2x2<bf16_t, 64, 4, true>(a, s);
    }
    if (cin == 256) return launch_convt2x2<float, 256, 2, true>(a, s);
    if (cin == 128) return launch_convt2x2<float, 128, 4, true>(a, s);
    return launch_convt2x2<float, 64, 4, true>(a, s);
  }
  if (prec == DFA_PREC_BF16) {
    if (cin == 256) return launch_convt2x2<bf16_t, 256, 4>(a, s);
    if (cin == 128) return launch_convt2x2<bf16_t, 128, 4>(a, s);
    return launch_convt2x2<bf16_t, 64, 4>(a, s);
  }
  if (cin == 256) return launch_convt2x2<float, 256, 2>(a, s);
  if (cin == 128) return launch_convt2x2<float, 128, 4>(a, s);
  return launch_convt2x2<float, 64, 4>(a, s);
}

// number of [COUT][2] records launch_cae_dec writes to a.stats_partial for P = B H W input pixels
int cae_dec_stats_records(int prec, int cin, long P) {
  return (prec != DFA_PREC_BF16 && cin == 256) ? convt2x2_blocks<2>(P) : convt2x2_blocks<4>(P);
}

}  // namespace dfa
