// conv3x3_inst_cnn2d.hip -- the conv3x3_mfma instantiations used by the CNN2D forward (src/model.py:21-29,37).
//   block 2: 32 -> 64 channels, AvgPool2d((2,1)) epilogue;   block 3: 64 -> 128 channels, mean-over-T epilogue.
// Template arguments <T, CIN, NSL, MG, RP, MT, EPI, MINW>: see conv3x3_mfma.h.  MINW = waves/SIMD the register
// allocator is asked to fit: the fp32 kernels keep 144/288 weight VGPRs per lane and run at one wave per SIMD.
#include "dfa_internal.h"

namespace dfa {

// dma = 1: stage the input ring with global_load_lds (LDS-DMA), 0: through registers, -1: the faster of the two as
// measured on MI355X at [256,321,180] (interleaved A/B, tools/gpu_ab.py) -- LDS-DMA everywhere: bf16 block 2 0.257 vs
// 0.285 ms, bf16 block 3 0.410 vs 0.420 ms, fp32 2.15 vs 2.34 ms and 4.14 vs 4.46 ms.
// pipe = 0 selects the compiler-scheduled twins of the asm-pipelined bf16 kernels (same arithmetic, bit-identical
// output): the GPU tests run both and compare.
hipError_t launch_cnn2d_block2(int prec, const ConvArgs& a, hipStream_t s, int dma, int pipe) {
  if (dma < 0 || dma >= 2) dma = 1;
  if (prec == DFA_PREC_BF16) {
    if (dma) return pipe ? launch_conv3x3<bf16_t, 32, 2, 2, 2, 1, EPI_POOL_H2, 2, false, true>(a, s)
                         : launch_conv3x3<bf16_t, 32, 2, 2, 2, 1, EPI_POOL_H2, 2, false, true, false, 0>(a, s);
    return pipe ? launch_conv3x3<bf16_t, 32, 2, 2, 2, 1, EPI_POOL_H2, 2>(a, s)
                : launch_conv3x3<bf16_t, 32, 2, 2, 2, 1, EPI_POOL_H2, 2, false, false, false, 0>(a, s);
  }
  if (dma) return launch_conv3x3<float, 32, 2, 2, 2, 1, EPI_POOL_H2, 1, false, true>(a, s);
  return launch_conv3x3<float, 32, 2, 2, 2, 1, EPI_POOL_H2, 1>(a, s);
}

hipError_t launch_cnn2d_block3(int prec, const ConvArgs& a, hipStream_t s, int dma, int pipe) {
  if (dma < 0 || dma >= 2) dma = 1;
  if (prec == DFA_PREC_BF16) {
    if (dma) return pipe ? launch_conv3x3<bf16_t, 64, 4, 1, 1, 1, EPI_MEAN_T, 2, false, true>(a, s)
                         : launch_conv3x3<bf16_t, 64, 4, 1, 1, 1, EPI_MEAN_T, 2, false, true, false, 0>(a, s);
    return pipe ? launch_conv3x3<bf16_t, 64, 4, 1, 1, 1, EPI_MEAN_T, 2>(a, s)
                : launch_conv3x3<bf16_t, 64, 4, 1, 1, 1, EPI_MEAN_T, 2, false, false, false, 0>(a, s);
  }
  if (dma) return launch_conv3x3<float, 64, 4, 1, 1, 1, EPI_MEAN_T, 1, false, true>(a, s);
  return launch_conv3x3<float, 64, 4, 1, 1, 1, EPI_MEAN_T, 1>(a, s);
}

}  // namespace dfa
