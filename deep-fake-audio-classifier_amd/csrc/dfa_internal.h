// dfa_internal.h -- context object and internal launch prototypes of libdfa_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/dfa_hip.h"
#include "conv3x3_mfma.h"
#include "rng.h"

namespace dfa {

constexpr float kBnEps = 1e-5f;  // torch.nn.BatchNorm default (src/model.py:16)

// ---- kernel timing slots (HIP events on the context's stream) -----------------------------------
constexpr int kMaxSlots = 16;
constexpr int kEventsPerSlot = 256;

struct SlotTimer {
  std::vector<hipEvent_t> start, stop;
  int used = 0;
};

struct PackedConv {
  uint4* wpack = nullptr;  // [COUT/32][9][CIN/KG][64]
  float* bias = nullptr;   // [COUT]
};

struct Cnn2dState {
  const float* p[DFA_CNN2D_NPARAMS] = {nullptr};
  bool have_params = false;
  int in_features = 0;
  int prepared_prec = -1;
  void* packed = nullptr;  // one device allocation holding everything below
  size_t packed_bytes = 0;
  float* w1 = nullptr;     // [32][9] folded
  float* b1 = nullptr;     // [32]
  uint4* c1pack = nullptr; // [4][64] bf16 hi/lo A operands of the fused block 1+2 kernel (conv12_fused.hip)
  float* c1bias = nullptr; // [32]
  uint4* c3_m16 = nullptr; // block 3 weights in the 16x16x32 operand order (conv3_m16.hip), bf16 mode only
  PackedConv c2, c3;
  // train mode (train_api.hip)
  void* train_packed = nullptr;
  float *tw1 = nullptr, *tb1 = nullptr;   // conv1 folded with the current batch statistics
  PackedConv t2, t3, d2, d3;              // raw forward images and data-gradient images
  DropCfg train_drop{};
  AugCfg aug_armed{};                     // dfa_cnn2d_set_train_augment: consumed by the next forward_train
  AugCfg train_aug{};                     // the augmentation of the forward_train in flight (its backward re-reads x through it)
  int train_prec = -1, train_B = 0, train_T = 0;
  int train_c1_fused = 0;                 // the forward left XX / Xs behind for the one-pass conv1 backward
  int train_c1_mfma = 0;                  // this step's block-1 passes ran on train_conv1_mfma.hip
  int train_dgrad_m16 = 0;                // the d2/d3 images are in the 16x16x32 order of conv_split.hip (bf16 mode)
};

// Synchronised BatchNorm for data-parallel training ("SyncBN via two small all-reduces per layer -- makes
// N GPUs x B equal to one GPU x N*B up to summation order"): between a layer's statistics reduction and its use the library copies
// the per-channel sums into `buf` and calls fn(user, buf, count) on the host thread; the callee enqueues an in-place SUM all-reduce
// of buf[0 .. count) ordered after the work already on the context's stream (torch.distributed does) and returns 0.
struct BnSync {
  int (*fn)(void* user, float* buf, int count) = nullptr;
  void* user = nullptr;
  int world = 1;
  float* buf = nullptr;   // caller-owned device buffer, >= 512 floats (2 x the widest layer's channels)
};
// local sums -> the sums the apply / finalize stage uses (buf after the all-reduce, or `sums` itself when not synchronising) and
// the factor that turns 1 / n_local into 1 / n_global
hipError_t bn_sync_sums(const BnSync* sy, const float* sums, int count, hipStream_t s, const float** sums_apply, float* inv_scale);

struct Cnn1dState {
  const float* p[DFA_CNN1D_NPARAMS] = {nullptr};
  bool have_params = false;
  bool prepared = false;
  int in_features = 0;
  void* packed = nullptr;
  float *w[3] = {nullptr, nullptr, nullptr}, *b[3] = {nullptr, nullptr, nullptr};
  float* wp[3] = {nullptr, nullptr, nullptr};   // MFMA A-fragment images of the folded weights (cnn1d_fused.hip)
  void* wx[3] = {nullptr, nullptr, nullptr};    // hi / lo bf16 A-fragment images (cnn1d_fused_x3.hip)
  // train mode: data-gradient weight images of conv layers 2 and 3 (+ a zero bias), dropout state
  void* train_packed = nullptr;
  float *wt[2] = {nullptr, nullptr}, *zero_bias = nullptr;
  int wx3_F = 0, train_x3 = 0;
  void* wx3[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};   // per-step bf16x3 A-fragment images: forward layers 1-3, data gradients 3->2, 2->1 (conv1d_x3_kernel)
  DropCfg train_drop{};
  int train_B = 0, train_T = 0;
  AugCfg aug_armed{};   // dfa_cnn1d_set_train_augment: consumed by the next forward_train
  AugCfg train_aug{};   // the augmentation of the forward_train in flight (the layer-1 weight gradient re-reads x through it)
};

struct CaeState {
  const float* p[DFA_CAE_NPARAMS] = {nullptr};
  bool have_params = false;
  int prepared_prec = -1;
  void* packed = nullptr;
  float *w1 = nullptr, *b1 = nullptr;
  PackedConv enc[3];   // encoder blocks 2-4
  PackedConv dec[3];   // decoder blocks 1-3 (ConvTranspose2d images)
  uint4* c1pack = nullptr;     // block-1 MFMA A operands [6][64]: three bf16 terms of the folded weights, even / odd row (cae_enc1_mfma.hip)
  float* c1bias = nullptr;     // [32] 0.5 * folded bias
  uint4* dec4pack = nullptr;   // decoder block 4 as MFMA A operands [4][64] (cae_dec_fused.hip)
  float* opad_cst = nullptr;   // [16] reconstruction constants of the output_padding columns (cae_dec_fused.hip)
  // train mode (cae_train_api.hip): raw forward images, data-gradient images, raw ConvTranspose2d images
  void* train_packed = nullptr;
  float *tw1 = nullptr, *tb1 = nullptr;
  int train_c1_mfma = 0;       // this step's block-1 passes run on the matrix cores (train_conv1_mfma.hip, 2x2-pool backward)
  int train_dgrad_m16 = 0;     // this step's 64 -> 32 and 128 -> 64 data-gradient images are in the 16x16x32 order of conv_split.hip (bf16 mode)
  int train_enc4_wide = 0;     // this step's block-4 forward / data-gradient images are 128-input-channel ones (bf16 mode, option cae_enc4_wide)
  PackedConv tenc[3], tdg[3], tdec[3];
  int train_prec = -1, train_B = 0, train_T = 0;
};

}  // namespace dfa

struct dfa_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  char err[512] = {0};
  void* zero_page = nullptr;   // 256 zero bytes: source of out-of-image chunks for LDS-DMA staging
  dfa::BnSync bn_sync;         // dfa_ctx_set_bn_sync: synchronised BatchNorm statistics in the three models' forward_train / backward
  int block3_m16 = 1;          // bf16 block 3 on v_mfma_f32_16x16x32_bf16 (conv3_m16.hip); 0 = the 32x32x16 kernel
  int fuse_conv1 = 1;          // bf16 mode: blocks 1 and 2 in one kernel (conv12_fused.hip; fp32 features are rounded to bf16 on load); 0 = two kernels
  int lds_pipe = 1;            // 1 = asm-pipelined LDS fragment reads where instantiated, 0 = compiler-scheduled twins (test hook)
  int time_split = -1;         // eval forward, small batches: -1 = automatic time-axis split, 0 = off, n > 0 = force n segments
  int conv1_bwd_fused = 1;     // CNN2D training: block-1 backward as ONE pass over da1 (train_conv1.hip BWD_FUSED); 0 = reduce pass + weight-gradient pass
  int conv1_mfma = 1;          // bf16 training, bf16 features, no folded augmentation: block-1 passes on the matrix cores (train_conv1_mfma.hip); 0 = vector-ALU kernels
  int dgrad_m16 = 1;           // bf16 training: data-gradient convolutions on the 16x16x32 kernel (conv_split.hip), one launch each; 0 = the 32x32x16 kernels
  int cae_enc1_mfma = 1;       // auto-encoder eval forward, bf16 mode: block 1 on the matrix cores (cae_enc1_mfma.hip); 0 = the vector-ALU kernel
  int cae_enc_dma = 1;         // auto-encoder eval forward, bf16 mode: encoder blocks 2-4 stage their input rows by LDS-DMA; 0 = through registers
  int cae_dgrad_mfma = 1;      // auto-encoder training, bf16 mode: ConvTranspose2d data gradients on the bf16 matrix cores writing bf16
  int cae_conv_stats = 1;      // auto-encoder training: encoder blocks 2-3 and decoder blocks 1-3 take their BatchNorm statistics in the convolution's epilogue (0 = separate pass over z)
  int cae_bwd_fold = 1;        // auto-encoder training: the decoder's BatchNorm-backward apply pass writes dz patch-major and sums the ConvTranspose2d bias gradient (0 = three passes)
  int cae_enc4_wide = 1;       // auto-encoder training, bf16 mode: encoder block 4 forward in ONE 128-input-channel launch, its data gradient in two (0 = 64-channel launches chained through fp32 partial sums)
                               // (convt_dgrad_bf16.hip); 0 = the fp32-MFMA GEMM + cast pass of round 2
  int cae_dec_fused = 1;       // auto-encoder eval forward, bf16 mode: decoder + squared error as ONE kernel (cae_dec_fused.hip); 0 = four launches
  int cnn1d_train_x3 = 1;      // CNN1D training convolutions (3 forward, 2 data gradients) on the matrix-core layer kernel (conv1d_x3_kernel) where
                               // its layout rules hold: 1 = three bf16 terms per operand (fp32-grade), 3 = two terms (bf16x3, ~1e-5: opt-in),
                               // 0 = the fp32 VALU kernel (conv1d.hip); 2 = diagnostic: as 1 with one channel tile per workgroup
  int cnn1d_fused = 1;         // CNN1D eval forward as ONE kernel when T <= 384: 1 = split-bf16 kernel (cnn1d_fused_x3.hip) for the reference's
                               // storage layout, the exact-fp32 one (cnn1d_fused.hip) otherwise; 2 = always the exact-fp32 one; 0 = the three-launch path
  int clock_probe = 0;         // 1 = the bf16 block-3 kernel stamps its main loop (s_memtime / s_memrealtime) into clock_buf: dfa_ctx_clock_read
  long long* clock_buf = nullptr;   // device, 1024 x {cycles, 100 MHz ticks}
  float* aug_keep = nullptr;        // device copy of the armed augmentation's keep mask (dfa_cnn2d_set_train_augment copies keep_f)
  int aug_keep_cap = 0, aug_keep_slot = 0;
  float* mse_partial = nullptr;     // device, kMseBlocks floats: block sums of dfa_mse_fwd_bwd (allocated on first use)
  int conv_dma = -1;           // conv input staging: 1 = global_load_lds (LDS-DMA), 0 = through registers, -1 = per-kernel default
  dfa::Cnn2dState cnn2d;
  dfa::Cnn1dState cnn1d;
  dfa::CaeState cae;
  unsigned timing = 0;         // bit s set: record HIP events around the launches of timing slot s
  dfa::SlotTimer slots[dfa::kMaxSlots];
};

namespace dfa {

inline int fail(dfa_ctx* ctx, int code, const char* fmt, ...) __attribute__((format(printf, 3, 4)));
inline int fail(dfa_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

#define DFA_HIP_CHECK(ctx, expr)                                                                  \
  do {                                                                                            \
    hipError_t e__ = (expr);                                                                      \
    if (e__ != hipSuccess)                                                                        \
      return dfa::fail(ctx, DFA_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
  } while (0)

// RAII-less helper: record start/stop events around a launch when ctx->timing is on
struct ScopedSlot {
  dfa_ctx* ctx;
  int slot;
  int idx;
  ScopedSlot(dfa_ctx* c, int s) : ctx(c), slot(s), idx(-1) {
    if (!((ctx->timing >> slot) & 1u)) return;
    SlotTimer& t = ctx->slots[slot];
    if (t.used >= kEventsPerSlot) return;
    if ((int)t.start.size() <= t.used) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      t.start.push_back(a);
      t.stop.push_back(b);
    }
    idx = t.used++;
    (void)hipEventRecord(t.start[idx], ctx->stream);
  }
  ~ScopedSlot() {
    if (idx >= 0) (void)hipEventRecord(ctx->slots[slot].stop[idx], ctx->stream);
  }
};

// ---- launch prototypes (one translation unit each, see Makefile) ---------------------------------
// pack.hip
hipError_t launch_fold_conv1(const float* w, const float* b, const float* g, const float* beta, const float* mean,
                             const float* var, float* w1, float* b1, int cout, hipStream_t s);
hipError_t launch_fold_pack_conv3x3(const float* w, const float* b, const float* g, const float* beta,
                                    const float* mean, const float* var, int cin_total, int cin_off, int cin, int cout,
                                    int prec, uint4* wpack, float* bias, hipStream_t s, int fold = 1,
                                    float post_scale = 1.0f);
hipError_t launch_pack_conv3x3_dgrad(const float* w, int cin, int cout, int co_off, int co_n, int prec, uint4* wpack,
                                     float* bias, hipStream_t s);
hipError_t launch_fold_pack_convt2x2(const float* w, const float* b, const float* g, const float* beta,
                                     const float* mean, const float* var, int cin, int cout, int prec, uint4* wpack,
                                     float* bias, hipStream_t s, int fold = 1);
// gemm_f32.hip
hipError_t launch_gemm_tn_bf16(const void* X, const void* Z, float* partial, size_t partial_floats, int M, int N, int K,
                               int* nparts, hipStream_t s);
hipError_t launch_gemm_f32(int a_bf16, const void* A, int64_t sam, int64_t sak, int b_bf16, const void* Bm, int64_t sbk,
                           int64_t sbn, float* C, int M, int N, int K, int ksplit, hipStream_t s);
// conv1.hip
hipError_t launch_conv1(const void* x, int x_dtype, int64_t sb, int64_t st, int64_t sf, const float* w1,
                        const float* b1, void* out, int out_prec, int B, int T, int F, hipStream_t s,
                        const DropCfg* drop = nullptr, const AugCfg* aug = nullptr);
// linear.hip
hipError_t launch_linear(const float* emb, const float* w, const float* bias, float* logits, int B, int K,
                         hipStream_t s);
hipError_t launch_emb_reduce(const float* parts, int nparts, size_t stride, size_t n, float inv_h, float* out, hipStream_t s);
// conv1d.hip
hipError_t launch_fold_conv1d(const float* w, const float* b, const float* g, const float* beta, const float* mean,
                              const float* var, float* wf, float* bf, int cin, int cout, hipStream_t s);
// cae_enc1_mfma.hip: auto-encoder block 1 (conv 1 -> 32 + ReLU + 2 x 2 pool) on the matrix cores, bf16 mode
hipError_t launch_cae_enc1_mfma(const void* x, int x_dtype, int64_t sb, int64_t st, int64_t sf, const float* mu, const float* sigma,
                                const uint4* c1pack, const float* c1bias, void* out, int B, int T, int F, hipStream_t s);
hipError_t launch_pack_cae_enc1_mfma(const float* w1, const float* b1, uint4* pack, float* bias, hipStream_t s);
// cae_dec_fused.hip: the auto-encoder's decoder + per-sample squared error as one kernel (bf16 mode)
int cae_dec_fused_tiles(int H4, int W4);
hipError_t launch_cae_opad_consts(const float* b2, const uint4* wp3, const float* b3, const float* w4, const float* b4, float* cst,
                                  hipStream_t s);
hipError_t launch_cae_dec_fused(const void* lat, const uint4* wp1, const float* b1, const uint4* wp2, const float* b2, const uint4* wp3,
                                const float* b3, const float* w4, const float* b4, const uint4* w4pack, const float* cst, const void* x,
                                int x_dtype, int64_t sb, int64_t st, int64_t sf, const float* mu, const float* sigma, float* recon,
                                float* partial, int B, int H4, int W4, int T, int F, hipStream_t s, long long* stamps = nullptr);
bool cae_dec_fused_supports(int T, int F, int64_t st, int64_t sf);
hipError_t launch_pack_cae_dec4(const float* w4, uint4* pack, hipStream_t s);
hipError_t launch_cae_mse_finalize(const float* partial, int nblk, float inv_n, float* mse, int B, hipStream_t s);
// cnn1d_fused.hip: the whole CNN1D eval forward as one kernel (fp32 matrix cores, activations in LDS)
int cnn1d_fused_ncp_pad(int cin, int layer);
size_t cnn1d_fused_pack_floats(int cin, int cout, int layer);
bool cnn1d_fused_supports(int T);
hipError_t launch_pack_cnn1d_fused(const float* wf, float* wp, int cin, int cout, int layer, hipStream_t s);
hipError_t launch_cnn1d_fused(const float* x, int64_t sb, int64_t st, int64_t sf, const float* wp1, const float* b1, const float* wp2,
                              const float* b2, const float* wp3, const float* b3, const float* cw, const float* cb, float* logits, int B,
                              int T, int F, hipStream_t s, long long* stamps = nullptr);
// cnn1d_fused_x3.hip: the same forward on the bf16 matrix cores with hi + lo bf16 operands (fp32-grade accuracy)
int cnn1d_x3_nks(int cin);
size_t cnn1d_x3_pack_bytes(int cin, int cout);
hipError_t launch_pack_cnn1d_x3(const float* wf, void* wx, int cin, int cout, hipStream_t s);
bool cnn1d_fused_x3_supports(const void* x, int64_t sb, int64_t st, int64_t sf, int T, int F);
hipError_t launch_cnn1d_fused_x3(const float* x, const void* w1, const float* b1, const void* w2, const float* b2, const void* w3,
                                 const float* b3, const float* cw, const float* cb, float* logits, int B, int T, int F, hipStream_t s,
                                 long long* stamps = nullptr);
hipError_t launch_conv1d(const float* x, int64_t sb, int64_t sc, int64_t st, const float* w, const float* bias,
                         float* out, int B, int Cin, int Cout, int T, bool mean, hipStream_t s, bool relu = true,
                         const AugCfg* aug = nullptr);
// train_cnn1d.hip
int cm_chunks(int B);
int conv1d_wgrad_chunks(int B);
hipError_t launch_cm_stats(const float* z, float* partial, int B, int C, int T, hipStream_t s);
hipError_t launch_cm_bn_relu_drop(const float* z, const float* mean, const float* invstd, const float* gamma,
                                  const float* beta, float* h, int B, int C, int T, const DropCfg& dc, hipStream_t s);
hipError_t launch_cm_bn_relu_meant(const float* z, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, float* pooled, int B, int C, int T, hipStream_t s);
hipError_t launch_cm_bn_bwd(int src, const float* z, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, const float* up, float* partial, float* sums, float* dz, int B, int C,
                            int T, const DropCfg& dc, hipStream_t s, const BnSync* sync = nullptr);
bool conv1d_x3_supports(const float* x, int64_t sb, int64_t sc, int64_t st, const float* z, int T, int Cin, int Cout, int terms);
hipError_t launch_conv1d_x3(const float* x, int64_t sb, const void* wx, const float* bias, float* z, int B, int Cin, int Cout, int T,
                            int terms, hipStream_t s, int mode = 1, const AugCfg* aug = nullptr);
size_t conv1d_terms_pack_bytes(int cin, int cout, int terms);
hipError_t launch_pack_conv1d_terms(const float* wf, void* wx, int cin, int cout, int terms, hipStream_t s);
hipError_t launch_pack_conv1d_train_all(const float* w1, const float* w2, const float* w3, void* const* dst, int F, int terms, float* zero_bias,
                                        hipStream_t s);
hipError_t launch_conv1d_wgrad(const float* dz, const float* h, int64_t hsb, int64_t hsc, int64_t hst, float* partial,
                               float* dw, float* db, int B, int Cin, int Cout, int T, hipStream_t s, const AugCfg* aug = nullptr, int x3 = 0);
hipError_t launch_conv1d_dgrad_pack(const float* w, float* wt, float* zero_bias, int cin, int cout, hipStream_t s);
// cae.hip
hipError_t launch_cae_enc1(const void* x, int x_dtype, int64_t sb, int64_t st, int64_t sf, const float* mu,
                           const float* sigma, const float* w1, const float* b1, void* out, int prec, int B, int T, int F,
                           hipStream_t s);
hipError_t launch_cae_opad_col(void* out, const float* bias, int prec, int rows, int Wo, int C, hipStream_t s,
                               int no_relu = 0);
int cae_dec4_blocks(int T, int W3);
hipError_t launch_cae_dec4_mse(const void* d3, int prec, const float* w4, const float* b4, const void* x, int x_dtype,
                               int64_t sb, int64_t st, int64_t sf, const float* mu, const float* sigma, float* recon,
                               float* partial, float* mse, int B, int H3, int W3, int T, int F, hipStream_t s);
hipError_t launch_cae_latent_export(const void* lat, int prec, float* out, int B, int HW, int C, hipStream_t s);
// train_elem.hip / train_conv1.hip / wgrad_mfma.hip
hipError_t launch_bn_finalize(const float* partial, int nparts, int C, double n, float* mean, float* var, float* invstd,
                              float* running_mean, float* running_var, float momentum, hipStream_t s);
hipError_t launch_reduce_partials(const float* partial, int nparts, int n, float scale, float* out, hipStream_t s,
                                  float* scratch = nullptr);
hipError_t launch_reduce_partials_strided(const float* partial, int nparts, int stride, int off, int n, float* out,
                                          hipStream_t s);
hipError_t launch_bn_relu_pool_drop(int prec, const void* z, const float* mean, const float* invstd, const float* gamma,
                                    const float* beta, void* out, int B, int H, int W, int C, const DropCfg& dc,
                                    hipStream_t s);
int cl_stats_blocks(size_t npix, int* pix_per_block);
hipError_t launch_cl_stats(int prec, const void* z, float* partial, size_t npix, int C, hipStream_t s);
hipError_t launch_bn_relu_pool(int prec, int pool, const void* z, const float* mean, const float* invstd,
                               const float* gamma, const float* beta, void* out, int B, int H, int W, int C,
                               hipStream_t s);
hipError_t launch_bn_relu_meant(int prec, const void* z, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, float* emb, int B, int H, int W, int C, hipStream_t s, float* msum = nullptr);
hipError_t launch_bn_bwd_meant_saved(int prec, const void* z, const float* mean, const float* invstd, const float* gamma,
                                     const float* beta, const float* demb, const float* msum, float* partial, float* sums,
                                     void* dz, int B, int H, int W, int C, hipStream_t s, const BnSync* sync = nullptr);
hipError_t launch_linear_bwd(const float* dlogits, const float* w, const float* emb, float* demb, float* dw, float* db,
                             int B, int K, hipStream_t s, int tc = 0, int tw = 0);
int bn_bwd_blocks(int B, int H, int W, int* pix_per_block);
// SRC_DIRECT apply pass that writes dz patch-major (the pixel-unshuffled operand of a ConvTranspose2d(k2, s2) layer's gradient GEMMs,
// Wh input columns) and leaves [nrec][C] records of its channel sums in bias_partial (train_elem.hip bn_bwd_apply_unshuffle_kernel)
struct BnBwdFold { void* zp; int Wh; float* bias_partial; int nrec; };
hipError_t launch_bn_bwd(int prec, int src, const void* z, const float* mean, const float* invstd, const float* gamma,
                         const float* beta, const float* demb, const void* da, float* partial, float* sums, void* dz,
                         int B, int H, int W, int C, const DropCfg& dc, hipStream_t s, float* scratch = nullptr, const BnSync* sync = nullptr,
                         BnBwdFold* fold = nullptr);
hipError_t launch_bce_smooth(const float* logits, const float* labels, float eps, int B, float* loss, float* dlogits,
                             hipStream_t s);
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                        float wd, int step, float grad_scale, hipStream_t s);
int conv1_train_blocks(int B, int T, int F);
// train_conv1_mfma.hip: the block-1 train passes on the matrix cores (bf16 features, no folded augmentation)
enum { C1X_STATS = 0, C1X_FWD = 1, C1X_BWD = 2 };
int conv1_mfma_blocks(int B, int T, int F);
hipError_t launch_conv1_mfma(int mode, const void* x, int64_t sb, int64_t st, int64_t sf, const float* w, const float* bias,
                             void* a1, const void* da1, float* partial, int B, int T, int F, const DropCfg& dc, hipStream_t s, int poolw = 1);
hipError_t launch_conv1_train(int mode, const void* x, int x_dtype, int64_t sb, int64_t st, int64_t sf, const float* w,
                              const float* bconv, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, const float* sums, const void* da1, int prec, float* partial, int B,
                              int T, int F, const DropCfg& dc, hipStream_t s, int poolw = 1, const AugCfg* aug = nullptr, float inv_n_scale = 1.0f);
hipError_t launch_conv1_bwd_finalize(const float* rec, const float* xxs, const float* w, const float* bconv, const float* mean,
                                     const float* invstd, const float* gamma, double n, float* dw, float* db, float* dgamma,
                                     float* dbeta, hipStream_t s, int derive_s2 = 0);
// cae_train.hip
hipError_t launch_pixel_unshuffle(int prec, const void* dz, void* zp, int B, int H, int W, int Wo, int C, hipStream_t s);
hipError_t launch_convt_w_to_q(const float* w, float* wq, int cin, int cout, hipStream_t s);
hipError_t launch_convt_q_to_w(const float* dwq, float* dw, int cin, int cout, hipStream_t s);
int cae_dec4_bwd_blocks();
// MSE form of the decoder-block-4 backward (drecon == nullptr): the upstream gradient 2 (recon - x) / (B T F) is formed in the kernel
struct MseArgs {
  const void* x;
  int x_bf16;
  int64_t sb, st, sf;
  const float* b4_dev;
};
hipError_t launch_cae_dec4_bwd(int prec, const void* d3, const float* w4, const float* drecon, void* dd3, float* partial,
                               int B, int H3, int W3, int T, int F, hipStream_t s, const MseArgs* mse_args = nullptr);
// loss = mean((recon - x)^2), drecon = 2 (recon - x) / n  (src/train_cae.py:67-68,203); partial: >= kMseBlocks floats of scratch
constexpr int kMseBlocks = 1024;
hipError_t launch_mse_fwd_bwd(const float* recon, const void* x, int x_bf16, int64_t sb, int64_t st, int64_t sf, int B, int T, int F,
                              float* partial, float* loss, float* drecon, hipStream_t s);
bool convt_dgrad_bf16_supports(int Cin, int Cout);
hipError_t launch_convt_dgrad_bf16(const void* zp, const float* wq, void* wfrag, void* dx, long P, int Cin, int Cout, hipStream_t s);
hipError_t launch_cast_from_f32(int prec, const float* src, void* dst, size_t n, hipStream_t s);
hipError_t launch_reduce_wgrad_window(const float* partial, int nparts, int stride, int cin, int cout, int cin_total,
                                      int ci_off, int co_off, float* dw, hipStream_t s);
hipError_t launch_wgrad3x3_window(int prec, int cin, int cout, int cin_total, int cout_total, int ci_off, int co_off,
                                  const void* dz, const void* a, float* partial, float* dw, float* db, int B, int H,
                                  int W, int nwg, hipStream_t s);
hipError_t launch_wgrad3x3(int prec, int cin, int cout, const void* dz, const void* a, float* partial, float* dw,
                           float* db, int B, int H, int W, int nwg, hipStream_t s);
// conv3x3_inst_*.hip
hipError_t launch_fold_pack_conv3x3_m16(const float* w, const float* b, const float* g, const float* beta, const float* mean,
                                        const float* var, int cin, int cout, uint4* wpack, hipStream_t s, int fold = 1);
hipError_t launch_train_fwd3_m16(const ConvArgs& a, hipStream_t s, int pipe = 1);
// conv_split.hip (DFA_PREC_BF16X3)
hipError_t launch_fold_pack_conv3x3_split(const float* w, const float* b, const float* g, const float* beta, const float* mean,
                                          const float* var, int cin, int cout, uint4* wpack, float* bias, float post_scale,
                                          hipStream_t s);
hipError_t launch_cnn2d_block2_split(const ConvArgs& a, hipStream_t s, int pipe = 1);
hipError_t launch_cnn2d_block3_split(const ConvArgs& a, hipStream_t s, int pipe = 1);
hipError_t launch_pack_conv3x3_dgrad_m16(const float* w, int cin, int cout, uint4* wpack, float* bias, hipStream_t s);
hipError_t launch_train_dgrad3_m16(const ConvArgs& a, hipStream_t s, int pipe = 1);
hipError_t launch_train_dgrad2_m16(const ConvArgs& a, hipStream_t s, int pipe = 1);
hipError_t launch_cnn2d_block3_m16(const ConvArgs& a, hipStream_t s, int pipe = 1);
hipError_t launch_reduce_wgrad_record(const float* partial, int nparts, int stride, int cin, int cout, int cin_total,
                                      int ci_off, int co_off, float* dw, float* db, hipStream_t s, int perm = 0);
void set_train_conv_variant(int v);
int train_conv_variant();   // conv3x3_inst_train.hip (process-wide test hook)
void set_wgrad_variant(int v);   // wgrad_mfma.hip: bf16 weight-gradient kernel selection (process-wide test hook)
hipError_t launch_pack_conv1_mfma(const float* w1, const float* b1, uint4* c1pack, float* c1bias, hipStream_t s);
hipError_t launch_conv12_fused(const void* x, int x_dtype, int64_t sb, int64_t st, int64_t sf, const uint4* c1pack, const float* c1bias,
                               const uint4* wpack2, const float* bias2, void* a2, int B, int T, int F, hipStream_t s, int pipe = 1,
                               int seg_iters = 0);
hipError_t launch_cnn2d_block2(int prec, const ConvArgs& a, hipStream_t s, int dma = -1, int pipe = 1);
hipError_t launch_cnn2d_block3(int prec, const ConvArgs& a, hipStream_t s, int dma = -1, int pipe = 1);
struct ConvTArgs;
hipError_t launch_cae_enc2(int prec, const ConvArgs& a, hipStream_t s, int pipe = 1, int dma = 0);
hipError_t launch_cae_enc3(int prec, const ConvArgs& a, hipStream_t s, int pipe = 1, int dma = 0);
hipError_t launch_cae_enc4(int prec, const ConvArgs& a, float* raw_tmp, hipStream_t s, int dma = 0);
hipError_t launch_cae_dec(int prec, int cin, const ConvTArgs& a, hipStream_t s);
int cae_dec_stats_records(int prec, int cin, long P);

}  // namespace dfa
