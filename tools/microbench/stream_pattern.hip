// stream_pattern.hip -- what HBM rate does the ACCESS PATTERN of the training convolutions allow, with no arithmetic at all?
// A workgroup owns (utterance, SW-column strip) and walks down H rows two at a time, like conv_split / conv3x3_mfma: per
// iteration it reads 2 rows x (SW + 2) pixels x PIN bytes and writes 2 rows x SW pixels x POUT bytes.  Prints GB/s for several
// strip widths at the shapes of the 32<->64-channel layers (B = 256, H = 160, W = 180).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/stream_pattern.hip -o /tmp/stream_pattern && /tmp/stream_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int NT>
__global__ __launch_bounds__(NT) void walk(const uint4* __restrict__ in, uint4* __restrict__ out, int H, int W, int SW, int pin16,
                                           int pout16, int nstrips) {
  const int b = blockIdx.x / nstrips, strip = blockIdx.x % nstrips;
  const int f0 = strip * SW;
  uint4 acc = make_uint4(0, 0, 0, 0);
  const int in_chunks = 2 * (SW + 2) * pin16, out_chunks = 2 * SW * pout16;
  for (int t = 0; t < H; t += 2) {
    for (int e = threadIdx.x; e < in_chunks; e += NT) {
      const int r = e / ((SW + 2) * pin16), rem = e - r * (SW + 2) * pin16;
      const int px = rem / pin16, c = rem - px * pin16;
      int f = f0 - 1 + px; f = f < 0 ? 0 : (f >= W ? W - 1 : f);
      const uint4 v = in[(((size_t)b * H + t + r) * W + f) * pin16 + c];
      acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    for (int e = threadIdx.x; e < out_chunks; e += NT) {
      const int r = e / (SW * pout16), rem = e - r * SW * pout16;
      const int px = rem / pout16, c = rem - px * pout16;
      if (f0 + px < W) out[(((size_t)b * H + t + r) * W + f0 + px) * pout16 + c] = acc;
    }
  }
}

int main() {
  const int B = 256, H = 160, W = 180;
  struct { const char* name; int pin, pout; } layers[] = {{"64ch in -> 32ch out (dz2 -> da1)", 128, 64}, {"32ch in -> 64ch out (a1 -> z2)", 64, 128}};
  for (auto& L : layers) {
    const size_t in_bytes = (size_t)B * H * W * L.pin, out_bytes = (size_t)B * H * W * L.pout;
    uint4 *in, *out;
    hipMalloc(&in, in_bytes); hipMalloc(&out, out_bytes);
    hipMemset(in, 1, in_bytes);
    const int widths[] = {30, 32, 60, 90, 180};
    for (int SW : widths) {
      const int nstrips = (W + SW - 1) / SW;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(walk<256>, dim3(B * nstrips), dim3(256), 0, 0, in, out, H, W, SW, L.pin / 16, L.pout / 16, nstrips);
      hipEventRecord(e0);
      const int N = 10;
      for (int rep = 0; rep < N; ++rep) hipLaunchKernelGGL(walk<256>, dim3(B * nstrips), dim3(256), 0, 0, in, out, H, W, SW, L.pin / 16, L.pout / 16, nstrips);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= N;
      printf("%-34s strip %3d (%2d workgroups/utt): %.3f ms  %.2f TB/s algorithmic\n", L.name, SW, nstrips, ms, (in_bytes + out_bytes) / ms * 1e-9);
    }
    hipFree(in); hipFree(out);
  }
  return 0;
}
