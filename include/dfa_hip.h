/*
 * dfa_hip.h -- C ABI of libdfa_hip.so: the MI355X (gfx950) implementation of the LFCC-classifier
 * hot path of kingdomseed/Deep-Fake-Audio-Classifier.
 *
 * The reference has no FFI of its own: its boundary for this path is the nn.Module surface of
 * src/model.py:5-42 (CNN2D), src/model_cnn1d.py:5-46 (CNN1D), src/model_cae.py:20-125
 * (ConvAutoencoder) plus the loss/optimiser calls of src/train.py:71-76,311-330.  Each entry point
 * below names the reference code it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer adds on the reference side.
 *
 * Conventions
 *   - plain C types only; every pointer named "device" is a HIP device pointer owned by the caller
 *     (PyTorch-ROCm tensors are used for storage only);
 *   - every function returns 0 (DFA_OK) or a negative DFA_E_* code; dfa_last_error() gives the text;
 *   - a context is bound to one device and one HIP stream; calls enqueue work on that stream and
 *     return without synchronising; a context is not thread-safe; one process per GPU for data parallel;
 *   - no allocation on the hot path: activations live in a caller-provided workspace
 *     (dfa_workspace_bytes), the library owns only the packed (BN-folded, MFMA-ordered) weights.
 */
#ifndef DFA_HIP_H
#define DFA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFA_VERSION 100 /* 0.1.0 */

/* error codes */
#define DFA_OK 0
#define DFA_E_BAD_SHAPE (-1)
#define DFA_E_BAD_DTYPE (-2)
#define DFA_E_NULL_PTR (-3)
#define DFA_E_NOT_PREPARED (-4)
#define DFA_E_HIP (-5)
#define DFA_E_WORKSPACE (-6)
#define DFA_E_UNSUPPORTED (-7)

/* element types of the input tensor x */
#define DFA_DTYPE_F32 0
#define DFA_DTYPE_BF16 1

/* arithmetic mode of the convolution stack (chosen at prepare time)
 *   DFA_PREC_F32  : fp32 storage, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) -- parity mode, logits within 1e-4
 *   DFA_PREC_BF16 : bf16 storage of weights/activations, fp32 accumulate (v_mfma_f32_32x32x16_bf16) -- throughput mode
 *   DFA_PREC_BF16X3 : CNN2D only.  Every weight and activation is carried as hi + lo bf16 (16 significant bits) and every
 *                   product is three bf16 MFMAs (hi*hi + lo*hi + hi*lo) accumulated in fp32 -- parity-grade (logits within
 *                   1e-4 of the reference like DFA_PREC_F32) at 3/16 of the fp32-MFMA cost */
#define DFA_PREC_F32 0
#define DFA_PREC_BF16 1
#define DFA_PREC_BF16X3 2

/* model ids for dfa_workspace_bytes */
#define DFA_MODEL_CNN2D 0
#define DFA_MODEL_CNN1D 1
#define DFA_MODEL_CAE 2

typedef struct dfa_ctx dfa_ctx;

/* ---- lifecycle ------------------------------------------------------------------------------ */
int dfa_version(void);
/* hip_stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL for the default stream */
int dfa_ctx_create(int device_id, void* hip_stream, dfa_ctx** out);
int dfa_ctx_destroy(dfa_ctx* ctx);
int dfa_ctx_set_stream(dfa_ctx* ctx, void* hip_stream);
/* tuning / test switches (all default to the fastest verified path; the alternatives exist so the tests can compare):
 *   "conv_dma"      1 = stage the MFMA convolutions' input rows with LDS-DMA (global_load_lds), 0 = through registers,
 *                   -1 (default) = per-kernel choice (also: environment variable DFA_CONV_DMA=0|1 before dfa_ctx_create)
 *   "lds_pipe"      1 (default) = asm-pipelined LDS fragment reads in the bf16 kernels that have them, 0 = their
 *                   compiler-scheduled twins (bit-identical results)
 *   "fuse_conv1"    1 (default) = CNN2D bf16 mode on bf16 features runs blocks 1+2 as one kernel, 0 = two kernels
 *   "block3_m16"    1 (default) = CNN2D bf16 block 3 on v_mfma_f32_16x16x32_bf16, 0 = the 32x32x16 kernel
 *   "time_split"    -1 (default) = CNN2D eval forward splits the time axis over workgroups when the batch alone cannot fill
 *                   the chip (B * strips below the resident-workgroup count, e.g. the reference's predict batch of 32), 0 =
 *                   never, n > 0 = force n segments (at most 4).  Logits and embeddings are bit-identical for every setting and
 *                   batch size: the time mean is always summed in the same canonical chunks
 *   "conv1_mfma"    1 (default) = bf16 training on bf16 features without a folded augmentation: the block-1 statistics,
 *                   forward and fused backward passes run on the matrix cores (train_conv1_mfma.hip); 0 = vector-ALU kernels.
 *                   May be cleared between dfa_cnn2d_forward_train and dfa_cnn2d_backward (both backward kernels read the
 *                   same forward state; used by the twin test).  The auto-encoder's training step (bf16 mode, bf16 features, even F)
 *                   obeys the same option: statistics + fused backward (2 x 2-pool form) on the same kernels, forward on
 *                   cae_enc1_mfma.hip; 0 = a statistics pass, the eval kernel and two vector-ALU backward passes
 *   "conv1_bwd_fused" 1 (default) = block-1 backward as one pass over da1 + algebra, 0 = reduce pass + weight-gradient pass
 *   "dgrad_m16"     1 (default) = bf16 training: each data-gradient convolution is ONE launch of the 16x16x32 kernel
 *                   (conv_split.hip), 0 = the 32x32x16 kernels (block 3 as two Cin-half launches through fp32 partial sums)
 *   "wgrad_variant" 3 (default) = pipelined bf16 weight-gradient kernel, 30 = its compiler-scheduled twin, 2 = the
 *                   earlier 4-wave kernel (process-wide)
 *   "train_conv_variant" 2 (default) = pipelined two-wave bf16 training convolutions where they fit, 0 = their
 *                   compiler-scheduled twins, 1 = the one-wave-per-SIMD instantiations (process-wide)
 *   "cae_enc1_mfma" 1 (default) = auto-encoder eval forward in bf16 mode: block 1 (conv 1 -> 32 + ReLU + 2 x 2 pool) on the matrix
 *                   cores with hi + lo bf16 operands; 0 = the vector-ALU kernel
 *   "cae_enc_dma"   1 (default) = auto-encoder eval forward in bf16 mode: encoder blocks 2-4 stage their input rows by LDS-DMA;
 *                   0 = through registers (bit-identical)
 *   "cae_dgrad_mfma" 1 (default) = auto-encoder training step in bf16 mode: the three ConvTranspose2d data gradients on the bf16
 *                   matrix cores, bf16 result written in place (convt_dgrad_bf16.hip); 0 = fp32-MFMA GEMM + cast pass
 *   "cae_conv_stats" 1 (default) = auto-encoder training step: encoder blocks 2-3 and decoder blocks 1-3 take their BatchNorm batch statistics in the
 *                   convolution's epilogue (fp32 sums of the outputs before they are rounded for storage, as the CNN2D's blocks 2 / 3);
 *                   0 = a separate statistics pass over the stored output
 *   "cae_bwd_fold"  1 (default) = auto-encoder training step: the BatchNorm-backward apply pass of decoder blocks 1-3 writes dz in the
 *                   patch-major order the ConvTranspose2d gradient GEMMs read and sums the bias gradient on the way; 0 = apply pass,
 *                   channel-sum pass and pixel-unshuffle pass (same values)
 *   "cae_enc4_wide" 1 (default) = auto-encoder training step in bf16 mode: encoder block 4 (128 -> 256) forward as ONE launch over all 128
 *                   input channels and its data gradient as two 128-channel launches; 0 = two / four 64-channel launches chained
 *                   through fp32 partial sums (same products, other summation order)
 *   "cae_dec_fused" 1 (default) = auto-encoder eval forward in bf16 mode: the four decoder blocks, the zero time padding and the
 *                   per-sample squared error run as ONE kernel with the intermediates in LDS / registers; 0 = four launches
 *   "cnn1d_fused"   1 (default) = CNN1D eval forward as ONE kernel for T <= 384 (all three Conv1d layers on the matrix cores with the
 *                   activations in LDS, the frame mean and the classifier in its epilogue): hi + lo bf16 operands, three bf16 MFMAs
 *                   per product (fp32-grade: logits within 1e-4 of the reference) when x is the reference's contiguous [B,F,T]
 *                   storage, the exact-fp32 matrix-core kernel for any other strides; 2 = always the exact-fp32 kernel; 0 = the
 *                   three-launch path
 *   "cnn1d_train_x3" 1 (default) = the five convolutions of a CNN1D training step (3 forward, 2 data gradients) on the matrix cores
 *                   (conv1d_x3_kernel) when the tensors are in the stored [B,F,T] layout, T <= 384, F % 4 == 0: every fp32 operand
 *                   as three bf16 terms (its 24-bit mantissa exactly), six MFMAs per product -- sums of fp32 grade; a folded
 *                   augmentation (dfa_cnn1d_set_train_augment) is applied as the input slabs are staged.  3 = two terms, three
 *                   MFMAs (bf16x3, ~1e-5 per product: gradients then carry ReLU-flip noise of ~3e-3 relative L2; opt-in);
 *                   0 = the fp32 vector-ALU kernels; 2 = diagnostic (as 1, one channel tile per workgroup)
 *   "clock_probe"   1 = the dominant kernel of the bf16 eval forward (CNN2D block 3) brackets its main loop with s_memtime /
 *                   s_memrealtime stamps (lane 0 of the first 1024 workgroups, into a buffer no kernel reads); dfa_ctx_clock_read
 *                   returns the shader clock the chip held inside that kernel.  0 (default) = two scalar compares per workgroup
 *   "poison_lds"    (action, test hook) fills all 160 KB of LDS of every CU with the 16-bit pattern `value` (0xffff / 0x7fc0 =
 *                   NaN, 0x7f80 = +Inf) on the context's stream.  LDS is not cleared between workgroups; the stale-LDS tests
 *                   (tests/test_lds_poison_gpu.py) run every path after this and require bit-identical results
 * unknown names return DFA_E_UNSUPPORTED */
int dfa_ctx_set_option(dfa_ctx* ctx, const char* name, int value);
const char* dfa_last_error(const dfa_ctx* ctx);
const char* dfa_error_name(int code);

/* ---- CNN2D (replaces CNN2D.__init__/forward, src/model.py:12-42) -------------------------------- */
/* params: 20 device pointers (fp32) in state_dict order without num_batches_tracked:
 *   conv.0.{weight,bias}, conv.1.{weight,bias,running_mean,running_var},
 *   conv.5.{weight,bias}, conv.6.{weight,bias,running_mean,running_var},
 *   conv.10.{weight,bias}, conv.11.{weight,bias,running_mean,running_var},
 *   classifier.{weight,bias}
 * The pointers are remembered (not copied); base_channels must be 32. */
#define DFA_CNN2D_NPARAMS 20
int dfa_cnn2d_set_params(dfa_ctx* ctx, const float* const* device_params, int n, int in_features,
                         int base_channels);
/* (re)build the BN-folded, MFMA-ordered weight images for eval-mode forward; call after any weight
 * change (load_state_dict, optimiser step) or precision switch. */
int dfa_cnn2d_prepare(dfa_ctx* ctx, int precision);
/* eval-mode forward: logits[b] = classifier(flatten(mean_T(conv(x[b]))))   (src/model.py:33-42)
 *   x: device, element (b,t,f) at x + b*stride_b + t*stride_t + f*stride_f (strides in elements; the
 *      reference harness passes the transposed view of the stored [B,F,T] tensor, src/predict.py:105);
 *   logits: device float[B]; embedding: device float[B*128*F] in c*F+f order, or NULL
 *      (src/model.py:40-41, src/embedding_anomaly.py:61);
 *   workspace: device, >= dfa_workspace_bytes(ctx, DFA_MODEL_CNN2D, B, T, F, precision) bytes. */
int dfa_cnn2d_forward(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b,
                      int64_t stride_t, int64_t stride_f, float* logits, float* embedding,
                      void* workspace, size_t workspace_bytes);

/* ---- CNN2D training step (replaces, for src/train.py:71-76, torch autograd over src/model.py:13-39) ------------ */
size_t dfa_cnn2d_train_workspace_bytes(const dfa_ctx* ctx, int B, int T, int F, int precision);
/* train-mode forward: BatchNorm uses batch statistics (and updates running_mean/var in place through the pointers
 * given to dfa_cnn2d_set_params when update_running_stats != 0, momentum 0.1 in the reference); Dropout(p_drop) after
 * pools 1 and 2 with a Philox mask keyed by (seed, offset).  Uses the raw parameters directly (no prepare needed).
 * The workspace keeps what dfa_cnn2d_backward needs; it must stay untouched until then. */
int dfa_cnn2d_forward_train(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b,
                            int64_t stride_t, int64_t stride_f, int precision, float p_drop, uint64_t seed,
                            uint64_t offset, float momentum, int update_running_stats, float* logits,
                            float* embedding, void* workspace, size_t workspace_bytes);
/* gradients of the 14 parameters w.r.t. sum_b dlogits[b]*logit[b], written (not accumulated) to grads[0..13] in
 * parameters() order: conv.0.{weight,bias}, conv.1.{weight,bias}, conv.5.*, conv.6.*, conv.10.*, conv.11.*,
 * classifier.{weight,bias}.  Pointing grads[] into ONE flat buffer gives the single all-reduce payload of data-parallel
 * training (464,644 bytes). */
int dfa_cnn2d_backward(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b,
                       int64_t stride_t, int64_t stride_f, const float* dlogits, float* const* grads, int ngrads,
                       void* workspace, size_t workspace_bytes);
/* loss = mean(BCEWithLogits(logits, y*(1-eps)+eps/2)), dlogits = dloss/dlogits   (src/train.py:311-320); device ptrs;
 * loss and dlogits may be NULL. */
int dfa_bce_smooth_fwd_bwd(dfa_ctx* ctx, const float* logits, const float* labels, float label_smoothing, int B,
                           float* loss, float* dlogits);
/* train-time feature augmentation in one pass (replaces the torch-op chain of src/augmentation.py:5-186 as composed by
 * src/train.py:68-69 / 271-289): out[b][t][f] = keep_f[f] * mask(x[b][(t - shift) mod T][f]) + jitter_std * N(0,1).
 * The caller draws the per-batch parameters exactly as the reference does (Python `random` spans and shift, torch
 * generator keep mask); mask spans refer to the frames / feature dims BEFORE the shift (the reference's op order:
 * time mask, feature mask, roll, channel drop, jitter); len = 0 disables a mask, keep_f = NULL and jitter_std = 0 disable
 * the other two; the noise is the library's Philox stream keyed by (seed, offset + (b*T + t)*F + f).  Out of place; any strides. */
int dfa_augment_batch(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b, int64_t stride_t,
                      int64_t stride_f, void* out, int out_dtype, int64_t out_stride_b, int64_t out_stride_t,
                      int64_t out_stride_f, int shift, const float* keep_f, int tmask_start, int tmask_len,
                      int fmask_start, int fmask_len, float jitter_std, uint64_t seed, uint64_t offset);
/* The same augmentation FOLDED INTO THE LOADS of the training step (src/train.py:68-69 applies the
 * augmentation to the batch and hands the result to the model): arms the parameters for the NEXT dfa_cnn2d_forward_train /
 * dfa_cnn2d_backward pair, whose three kernels that read x (statistics pass, block 1, block-1 backward) then read
 * keep_f[f] * mask(x[b][(t - shift) mod T][f]) + jitter_std * N(0,1) instead of x -- no augmented copy of the batch is
 * written or re-read.  One-shot: consumed by that forward (also by one that fails its argument checks); enable = 0 disarms.
 * keep_f (device float[F] or NULL) is copied into a context-owned buffer on the stream: the caller's tensor may be released at
 * once.  The element formula, noise stream included, is the one of dfa_augment_batch. */
int dfa_cnn2d_set_train_augment(dfa_ctx* ctx, int enable, int T, int F, int shift, const float* keep_f, int tmask_start,
                                int tmask_len, int fmask_start, int fmask_len, float jitter_std, uint64_t seed,
                                uint64_t offset);
/* The same for the CNN1D (src/train.py:68-69 feeds both classifiers): arms the augmentation for the NEXT dfa_cnn1d_forward_train /
 * dfa_cnn1d_backward pair, whose two kernels that read x (layer-1 convolution, layer-1 weight gradient) read it through the same
 * element formula -- the stand-alone dfa_augment_batch pass (one read + one write of the batch) disappears.  Same one-shot and
 * keep_f rules as dfa_cnn2d_set_train_augment. */
int dfa_cnn1d_set_train_augment(dfa_ctx* ctx, int enable, int T, int F, int shift, const float* keep_f, int tmask_start,
                                int tmask_len, int fmask_start, int fmask_len, float jitter_std, uint64_t seed,
                                uint64_t offset);
/* Synchronised BatchNorm for data-parallel training of all three models (the optional `dfa_bn_stats_{get,set}` pair, as
 * ONE hook): the reference trains on one GPU with plain BatchNorm2d (src/model.py:16,22,28); N ranks x B utterances reproduce one
 * rank x N*B -- statistics, running statistics and gradients, up to summation order -- when every BatchNorm layer's per-channel sums
 * are added over the ranks between their reduction and their use: (sum x, sum x^2) in the forward, (sum dy, sum dy*xhat) in the
 * backward, 2*C floats each, six exchanges per step for the classifiers, fourteen for the auto-encoder.  With a hook set, the
 * forward_train / backward entry points copy those sums into `buf` (caller-owned DEVICE memory, >= 512 floats) and call fn(user, buf, count) on the calling thread; fn enqueues an in-place
 * SUM all-reduce of buf[0 .. count) ordered after the work already on the context's stream (torch.distributed.all_reduce on the
 * tensor that owns buf does) and returns 0.  dgamma / dbeta stay the rank's own sums: the flat-gradient all-reduce adds them.
 * Block 1 of the CNN2D / auto-encoder then runs the two-pass vector backward (the one-pass moment algebra needs the sums too late).
 * fn = NULL: local statistics (torch DistributedDataParallel's default; what dfa_amd.train / train_cae use unless --sync-bn is given).
 * Set it before a forward_train; it stays in force until changed. */
typedef int (*dfa_bn_sync_fn)(void* user, float* buf, int count);
int dfa_ctx_set_bn_sync(dfa_ctx* ctx, dfa_bn_sync_fn fn, void* user, int world, float* buf, int capacity);
/* one torch.optim.AdamW step over a flat fp32 buffer: p *= 1-lr*wd; m,v update; p -= lr/bc1 * m/(sqrt(v)/sqrt(bc2)+eps).
 * grad_scale multiplies the gradient first (1/world after a sum all-reduce).  step is 1-based. */
int dfa_adamw_step(dfa_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale);

/* ---- CNN1D (replaces CNN1D.forward, src/model_cnn1d.py:37-46) -------------------------------------- */
/* params: 20 device pointers (fp32) in state_dict order without num_batches_tracked:
 *   conv.0.{weight,bias}, conv.1.{weight,bias,running_mean,running_var}, conv.4.*, conv.5.*, conv.8.*, conv.9.*,
 *   classifier.{weight,bias}.   base_channels must be 32; in_features is the Conv1d input-channel count. */
#define DFA_CNN1D_NPARAMS 20
int dfa_cnn1d_set_params(dfa_ctx* ctx, const float* const* device_params, int n, int in_features,
                         int base_channels);
int dfa_cnn1d_prepare(dfa_ctx* ctx);
/* eval-mode forward, fp32 arithmetic.  x as in dfa_cnn2d_forward (fp32 only); logits: device float[B]. */
int dfa_cnn1d_forward(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b,
                      int64_t stride_t, int64_t stride_f, float* logits, void* workspace,
                      size_t workspace_bytes);

/* CNN1D training step (fp32): same contract as the CNN2D pair above; dropout follows ReLU in blocks 1 and 2
 * (src/model_cnn1d.py:20,26); grads[0..13] in parameters() order conv.0.*, conv.1.*, conv.4.*, conv.5.*, conv.8.*, conv.9.*,
 * classifier.*. */
size_t dfa_cnn1d_train_workspace_bytes(const dfa_ctx* ctx, int B, int T, int F);
int dfa_cnn1d_forward_train(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b,
                            int64_t stride_t, int64_t stride_f, float p_drop, uint64_t seed, uint64_t offset,
                            float momentum, int update_running_stats, float* logits, void* workspace,
                            size_t workspace_bytes);
int dfa_cnn1d_backward(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b,
                       int64_t stride_t, int64_t stride_f, const float* dlogits, float* const* grads, int ngrads,
                       void* workspace, size_t workspace_bytes);

/* ---- ConvAutoencoder (replaces ConvAutoencoder.forward, src/model_cae.py:83-125, and the per-sample MSE of
 *      src/evaluation_cae.py:52-53 / src/hybrid_ensemble.py:55) ------------------------------------------------ */
/* params: 44 device pointers (fp32) in state_dict order without num_batches_tracked:
 *   encoder.{0,4,8,12}.{weight,bias} each followed by encoder.{1,5,9,13}.{weight,bias,running_mean,running_var};
 *   decoder.{0,3,6}.{weight,bias} each followed by decoder.{1,4,7}.{weight,bias,running_mean,running_var};
 *   decoder.9.{weight,bias}.   ConvTranspose2d weights keep torch's (Cin, Cout, 2, 2) layout.  base_channels = 32. */
#define DFA_CAE_NPARAMS 44
int dfa_cae_set_params(dfa_ctx* ctx, const float* const* device_params, int n, int base_channels);
int dfa_cae_prepare(dfa_ctx* ctx, int precision);
/* eval-mode forward.  x as in dfa_cnn2d_forward; T >= 16; F must satisfy F == 16*(F/16) + 4 (180 does), because
 * the decoder's output_padding=(0,1) is fixed (src/model_cae.py:68-69).
 *   mu, sigma: device float[F] or both NULL.  When given, the FeatureNormalizer z-score (x - mu[f]) / sigma[f]
 *              (src/dataset_cae.py:37-41) is fused into every read of x, i.e. x is the RAW feature tensor;
 *   recon:  device float[B*T*F] or NULL  -- reconstruction, rows >= 16*(T/16) are zero (model_cae.py:116-119);
 *   latent: device float[B*256*(T/16)*(F/16)] (NCHW) or NULL;
 *   mse:    device float[B] or NULL      -- mean over T*F of (recon - x_normalised)^2. */
int dfa_cae_forward(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b,
                    int64_t stride_t, int64_t stride_f, const float* mu, const float* sigma, float* recon,
                    float* latent, float* mse, void* workspace, size_t workspace_bytes);

/* ConvAutoencoder training step (replaces, for src/train_cae.py:58-82, torch autograd over src/model_cae.py:32-125).
 * forward_train: x is the (already z-scored) input; BatchNorm uses batch statistics and updates the running statistics
 * in place when update_running_stats != 0; recon, latent and mse may each be NULL (at least one of recon / mse is required;
 * mean_b mse[b] is MSELoss(recon, x) of src/train_cae.py:67-68, every sample having T*F elements).
 * backward: drecon = d(loss)/d(reconstruction), device float[B*T*F] -- or NULL for the reference's own loss,
 * MSELoss(recon, x) (src/train_cae.py:67-68,203): its gradient 2*(recon-x)/(B*T*F) is then formed INSIDE the decoder's last
 * backward kernel from the saved activations and x, so neither the reconstruction nor its gradient has to exist in memory.
 * Gradients of the 30 parameters are written to grads[] in parameters() order: encoder.{0,1,4,5,8,9,12,13}.{weight,bias},
 * decoder.{0,1,3,4,6,7}.{weight,bias}, decoder.9.{weight,bias}; pointing grads[] into ONE flat buffer gives the single
 * all-reduce payload of data-parallel training (2,246,532 bytes). */
size_t dfa_cae_train_workspace_bytes(const dfa_ctx* ctx, int B, int T, int F, int precision);
int dfa_cae_forward_train(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b,
                          int64_t stride_t, int64_t stride_f, int precision, float momentum,
                          int update_running_stats, float* recon, float* latent, float* mse, void* workspace,
                          size_t workspace_bytes);
int dfa_cae_backward(dfa_ctx* ctx, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b, int64_t stride_t,
                     int64_t stride_f, const float* drecon, float* const* grads, int ngrads, void* workspace,
                     size_t workspace_bytes);

/* MSELoss(recon, x) forward + backward in one pass (replaces nn.MSELoss + its autograd node, src/train_cae.py:67-68,203):
 * loss[0] = mean((recon - x)^2) over B*T*F elements (fixed-order two-stage reduction), drecon = 2*(recon - x)/(B*T*F).
 * recon: device float[B*T*F] contiguous; x: as in dfa_cae_forward (dtype + element strides); loss / drecon: device, either may be
 * NULL.  For callers that keep an explicit reconstruction; the native trainer uses dfa_cae_backward(drecon = NULL) instead. */
int dfa_mse_fwd_bwd(dfa_ctx* ctx, const float* recon, const void* x, int x_dtype, int B, int T, int F, int64_t stride_b,
                    int64_t stride_t, int64_t stride_f, float* loss, float* drecon);

/* ---- shared ------------------------------------------------------------------------------------ */
size_t dfa_workspace_bytes(const dfa_ctx* ctx, int model, int B, int T, int F, int precision);
/* names of the device kernels a forward launches, for profile post-processing ("" when unknown) */
const char* dfa_dominant_kernel(int model, int precision);
/* ---- per-kernel timing (HIP events recorded on the context's stream around every launch) -------------
 * slots for CNN2D: 0 = conv1, 1 = block 2 (MFMA), 2 = block 3 (MFMA, the dominant kernel), 3 = linear;
 * CNN1D: 4 = the fused forward (three-launch path: 4, 5, 6 = conv blocks, 7 = linear);  CAE: 8 = enc1, 9-11 = enc2-4 (MFMA), 12 = fused decoder + MSE (bf16 mode; four-launch path: 12-14 = dec1-3, 15 = dec4+MSE).
 * enable: 0 = off, 1 = all slots, any other value = bit mask of slots (e.g. 1<<2 = block 3 only).
 * Enable, run forwards, then read (read synchronises on the recorded events; call it outside timed regions).
 * At most 256 launches per slot are recorded between resets. */
int dfa_ctx_timing_enable(dfa_ctx* ctx, int enable);
int dfa_ctx_timing_reset(dfa_ctx* ctx);
int dfa_ctx_timing_read(dfa_ctx* ctx, int slot, float* total_ms, int* count);
/* The shader clock (GHz) the chip held inside the LAST bf16 CNN2D block-3 launch made with option "clock_probe" = 1: per
 * workgroup delta s_memtime / (delta s_memrealtime * 10 ns), median / min / max over the workgroups that stamped (at most
 * 1024).  Synchronises on the context's stream.  The MFMA peak is quoted at 2.4 GHz; under MFMA load the chip holds less
 * (MI355X_MICROARCH.md, DVFS give-back), so roofline fractions are reported both against the nominal peak and at this clock.
 * ghz_min / ghz_max may be NULL. */
int dfa_ctx_clock_read(dfa_ctx* ctx, double* ghz_median, double* ghz_min, double* ghz_max, int* workgroups);
/* Raw copy of the first n (<= 2048) 64-bit words of the probe buffer the "clock_probe" kernels stamp (diagnostics: the fused CNN1D
 * kernel writes, per workgroup b < 128, words 8b..8b+3 = s_memtime at start / after layer 1 / after layer 2 / at the end and
 * 8b+5, 8b+6 = s_memrealtime at start / end).  Synchronises on the context's stream. */
int dfa_ctx_debug_read(dfa_ctx* ctx, long long* host_words, int n);

#ifdef __cplusplus
}
#endif
#endif /* DFA_HIP_H */
