// conv3x3_inst_train.hip -- conv3x3_mfma instantiations of the CNN2D training step (src/train.py:71-76):
//   forward blocks 2/3 store the pre-BatchNorm output z (+ per-channel statistics partials);
//   data gradients:  da2 = conv3x3(dz3, W3') (128 -> 64 channels),  da1 = conv3x3(dz2, W2') (64 -> 32 channels).
// fp32 with 128 input channels is split over Cin exactly like the CAE encoder block 4 (conv3x3_inst_cae.hip).
#include "dfa_internal.h"

namespace dfa {

// The bf16 kernels that fit two waves per SIMD without scratch run with asm-pipelined LDS reads (conv3x3_mfma.h); variant
// 0 selects their compiler-scheduled twins (bit-identical; GPU test), 1 the earlier one-wave-per-SIMD instantiations.
static int g_train_conv_variant = 2;
void set_train_conv_variant(int v) { g_train_conv_variant = v; }
int train_conv_variant() { return g_train_conv_variant; }

hipError_t launch_train_fwd2(int prec, const ConvArgs& a, hipStream_t s) {
  if (prec == DFA_PREC_BF16) {
    if (g_train_conv_variant == 2) return launch_conv3x3<bf16_t, 32, 2, 2, 2, 1, EPI_PLAIN, 2, false, true, true, 4>(a, s);
    if (g_train_conv_variant == 0) return launch_conv3x3<bf16_t, 32, 2, 2, 2, 1, EPI_PLAIN, 2, false, true, true, 0>(a, s);
    return launch_conv3x3<bf16_t, 32, 2, 2, 2, 1, EPI_PLAIN, 1, false, false, true>(a, s);
  }
  return launch_conv3x3<float, 32, 2, 2, 2, 1, EPI_PLAIN, 1, false, false, true>(a, s);
}

hipError_t launch_train_fwd3(int prec, const ConvArgs& a, hipStream_t s) {
  if (prec == DFA_PREC_BF16) return launch_conv3x3<bf16_t, 64, 4, 1, 1, 1, EPI_PLAIN, 1, false, false, true>(a, s);
  return launch_conv3x3<float, 64, 4, 1, 1, 1, EPI_PLAIN, 1, false, true, true>(a, s);
}

// a.wpack: two consecutive [64/32][9][64/KG][64] images (Cin halves 0-63, 64-127).  Cin = 128 in ONE launch needs
// 288 weight VGPRs + the two-row epilogue and spills (1.44 ms measured for the bf16 form); two 64-channel launches
// (raw fp32 partial sums, then accumulate + store) run spill-free.
hipError_t launch_train_dgrad3(int prec, const ConvArgs& a, float* raw_tmp, hipStream_t s) {
  if (prec == DFA_PREC_BF16) {
    ConvArgs p1 = a;
    p1.in_pix_bytes = 128 * 2;
    p1.in_ch_off_bytes = 0;
    p1.raw_out = raw_tmp;
    hipError_t e = g_train_conv_variant == 2   ? launch_conv3x3<bf16_t, 64, 2, 2, 2, 1, EPI_RAW, 2, false, true, false, 3>(p1, s)
                   : g_train_conv_variant == 0 ? launch_conv3x3<bf16_t, 64, 2, 2, 2, 1, EPI_RAW, 2, false, true, false, 0>(p1, s)
                                               : launch_conv3x3<bf16_t, 64, 2, 2, 2, 1, EPI_RAW, 1, false, true>(p1, s);
    if (e != hipSuccess) return e;
    ConvArgs p2 = a;
    p2.in_pix_bytes = 128 * 2;
    p2.in_ch_off_bytes = 64 * 2;
    p2.acc_in = raw_tmp;
    p2.wpack = a.wpack + (size_t)(64 / 32) * 9 * 4 * 64;
    return launch_conv3x3<bf16_t, 64, 2, 2, 2, 1, EPI_PLAIN, 1, true, true>(p2, s);
  }
  ConvArgs p1 = a;
  p1.in_pix_bytes = 128 * 4;
  p1.in_ch_off_bytes = 0;
  p1.raw_out = raw_tmp;
  hipError_t e = launch_conv3x3<float, 64, 2, 2, 2, 1, EPI_RAW, 1, false, true>(p1, s);
  if (e != hipSuccess) return e;
  ConvArgs p2 = a;
  p2.in_pix_bytes = 128 * 4;
  p2.in_ch_off_bytes = 64 * 4;
  p2.acc_in = raw_tmp;
  p2.wpack = a.wpack + (size_t)(64 / 32) * 9 * 8 * 64;
  // variant 7 (diagnostic, tools/gpu_accin_probe.py): the asm-pipelined form of this one-wave-per-SIMD fp32 ACCIN kernel --
  // the configuration whose wrong sums round 1 recorded; never the default
  if (g_train_conv_variant == 7) return launch_conv3x3<float, 64, 2, 2, 2, 1, EPI_PLAIN, 1, true, true, false, 3>(p2, s);
  return launch_conv3x3<float, 64, 2, 2, 2, 1, EPI_PLAIN, 1, true, true>(p2, s);
}

hipError_t launch_train_dgrad2(int prec, const ConvArgs& a, hipStream_t s) {
  if (prec == DFA_PREC_BF16) return launch_conv3x3<bf16_t, 64, 1, 2, 2, 1, EPI_PLAIN, 1>(a, s);
  return launch_conv3x3<float, 64, 1, 2, 2, 1, EPI_PLAIN, 1, false, true>(a, s);
}

}  // namespace dfa
