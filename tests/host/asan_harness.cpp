// asan_harness.cpp -- drives the HOST side of the C ABI (include/dfa_hip.h) under AddressSanitizer.  Linked against
// libdfa_hip_asan.so, the host-only build of csrc/*.hip (`make -C deep-fake-audio-classifier_amd/csrc asan`: hipcc
// --offload-host-only -fsanitize=address; the kernels are absent, nothing is launched).  What runs without a device: every
// workspace planner over a sweep of shapes (pure host arithmetic), the error-name table, option parsing on a null context and
// every entry point's null-context / context-creation failure path.  With a device (the GPU box) a context is created as well
// and the argument checks that precede any launch are walked (bad dtype, bad shapes, short workspaces, unprepared models).
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "dfa_hip.h"

static int g_fail = 0;
#define EXPECT(cond)                                                         \
  do {                                                                       \
    if (!(cond)) { fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++g_fail; } \
  } while (0)

int main() {
  EXPECT(dfa_version() > 0);
  // ---- planners: multiples of 256, zero for impossible shapes; the training plans grow with B (the eval plan need not: small
  //      batches carry the extra embedding slabs of the time-axis split)
  const int Ts[] = {1, 3, 4, 5, 16, 37, 64, 130, 321, 700, 4001};
  const int Fs[] = {1, 16, 20, 33, 40, 65, 180, 224, 225, 512};
  const int Bs[] = {1, 2, 3, 16, 85, 256, 1024};
  for (int model = 0; model < 4; ++model)
    for (int prec = 0; prec < 4; ++prec)
      for (int T : Ts)
        for (int F : Fs) {
          for (int B : Bs) {
            const size_t n = dfa_workspace_bytes(nullptr, model, B, T, F, prec);
            EXPECT(n % 256 == 0);
          }
          EXPECT(dfa_workspace_bytes(nullptr, model, 0, T, F, prec) == 0);
        }
  for (int prec = 0; prec < 2; ++prec)
    for (int T : Ts)
      for (int F : Fs) {
        size_t p2 = 0, pc = 0, p1 = 0;
        for (int B : Bs) {
          const size_t a = dfa_cnn2d_train_workspace_bytes(nullptr, B, T, F, prec);
          const size_t c = dfa_cae_train_workspace_bytes(nullptr, B, T, F, prec);
          const size_t d = dfa_cnn1d_train_workspace_bytes(nullptr, B, T, F);
          EXPECT(a % 256 == 0 && c % 256 == 0 && d % 256 == 0);
          EXPECT(a >= p2 && c >= pc && d >= p1);
          EXPECT((T >= 4) == (a > 0));
          p2 = a; pc = c; p1 = d;
        }
      }
  // ---- tables and null-context paths
  for (int code = 1; code > -12; --code) EXPECT(dfa_error_name(code) != nullptr && strlen(dfa_error_name(code)) > 0);
  for (int model = -1; model < 5; ++model)
    for (int prec = -1; prec < 4; ++prec) EXPECT(dfa_dominant_kernel(model, prec) != nullptr);
  EXPECT(dfa_last_error(nullptr) != nullptr);
  EXPECT(dfa_ctx_set_option(nullptr, "time_split", 1) == DFA_E_NULL_PTR);
  EXPECT(dfa_ctx_destroy(nullptr) == DFA_E_NULL_PTR);
  EXPECT(dfa_ctx_create(0, nullptr, nullptr) == DFA_E_NULL_PTR);
  float out[4] = {0, 0, 0, 0};
  EXPECT(dfa_cnn2d_forward(nullptr, out, DFA_DTYPE_F32, 1, 16, 180, 2880, 180, 1, out, nullptr, out, 1024) == DFA_E_NULL_PTR);
  EXPECT(dfa_cnn2d_forward_train(nullptr, out, DFA_DTYPE_F32, 1, 16, 180, 2880, 180, 1, DFA_PREC_F32, 0.f, 0, 0, 0.1f, 1, out, nullptr, out, 1024) == DFA_E_NULL_PTR);
  EXPECT(dfa_adamw_step(nullptr, out, out, out, out, 4, 1e-3f, 0.9f, 0.999f, 1e-8f, 0.01f, 1, 1.f) == DFA_E_NULL_PTR);

  // ---- with a device: the argument checks in front of the launches
  dfa_ctx* ctx = nullptr;
  const int rc = dfa_ctx_create(0, nullptr, &ctx);
  if (rc != DFA_OK) {
    EXPECT(ctx == nullptr);
    EXPECT(dfa_ctx_create(-1, nullptr, &ctx) == DFA_E_HIP && ctx == nullptr);
    EXPECT(dfa_ctx_create(1 << 20, nullptr, &ctx) == DFA_E_HIP && ctx == nullptr);
    printf("asan harness: no device, host paths only, %d failure(s)\n", g_fail);
    return g_fail ? 1 : 0;
  }
  EXPECT(dfa_ctx_set_option(ctx, "no_such_option", 1) != DFA_OK);
  EXPECT(dfa_ctx_set_option(ctx, "time_split", 3) == DFA_OK);
  EXPECT(dfa_ctx_set_option(ctx, nullptr, 3) != DFA_OK);
  EXPECT(dfa_cnn2d_prepare(ctx, DFA_PREC_BF16) == DFA_E_NOT_PREPARED);
  EXPECT(strlen(dfa_last_error(ctx)) > 0);
  EXPECT(dfa_cnn2d_forward(ctx, out, DFA_DTYPE_F32, 1, 16, 180, 2880, 180, 1, out, nullptr, out, 1024) == DFA_E_NOT_PREPARED);
  EXPECT(dfa_cnn2d_forward_train(ctx, out, DFA_DTYPE_F32, 1, 16, 180, 2880, 180, 1, DFA_PREC_F32, 0.f, 0, 0, 0.1f, 1, out, nullptr, out, 1024) == DFA_E_NOT_PREPARED);
  EXPECT(dfa_cnn2d_set_params(ctx, nullptr, 20, 180, 32) == DFA_E_NULL_PTR);
  const float* ptrs[20];
  for (int i = 0; i < 20; ++i) ptrs[i] = out;
  EXPECT(dfa_cnn2d_set_params(ctx, ptrs, 19, 180, 32) != DFA_OK);
  EXPECT(dfa_cnn2d_set_params(ctx, ptrs, 20, 180, 31) != DFA_OK);
  EXPECT(dfa_cnn2d_set_train_augment(ctx, 1, 321, 180, 5, nullptr, 300, 40, 0, 0, 0.f, 1, 0) == DFA_E_BAD_SHAPE);
  EXPECT(dfa_bce_smooth_fwd_bwd(ctx, out, out, 0.7f, 4, out, out) == DFA_E_BAD_SHAPE);
  float ms = 0.f; int cnt = 0;
  EXPECT(dfa_ctx_timing_read(ctx, 99, &ms, &cnt) != DFA_OK);
  EXPECT(dfa_ctx_destroy(ctx) == DFA_OK);
  printf("asan harness: device present, %d failure(s)\n", g_fail);
  return g_fail ? 1 : 0;
}
