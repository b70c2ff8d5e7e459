// Does the 256 MB Infinity Cache carry the tail of one streaming pass over a 387 MB image into the next pass?
// Pass = 208 blocks x 512 threads reading the buffer once with 16-byte nontemporal loads, 32 KB contiguous per block and
// step (the projection kernel's access shape). "same" = every pass ascending; "flip" = passes alternate direction, so a
// pass starts where the previous one ended.   hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void pass_kernel(const floatx4* __restrict__ buf, long n_blocks32k, int reverse, int nt,
                                                   float* sink) {
  floatx4 acc = {0.f, 0.f, 0.f, 0.f};
  const long per = (n_blocks32k + gridDim.x - 1) / gridDim.x;
  // block b walks its own contiguous range (stream-K style); reverse: the ranges and the walk are mirrored
  for (long i = 0; i < per; ++i) {
    long u = (long)blockIdx.x * per + i;
    if (u >= n_blocks32k) break;
    if (reverse) u = n_blocks32k - 1 - u;
    const floatx4* p = buf + u * 2048 + threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      floatx4 v = nt ? __builtin_nontemporal_load(p + q * 512) : p[q * 512];
      acc += v;
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}
int main() {
  float* sink; CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long sizes_mb[] = {128, 200, 387, 774};
  for (long mb : sizes_mb) {
    const long nb = mb * 1024 * 1024 / 32768;
    floatx4* buf; CK(hipMalloc(&buf, nb * 32768)); CK(hipMemset(buf, 0, nb * 32768));
    float* other; CK(hipMalloc(&other, 64l << 20)); CK(hipMemset(other, 0, 64l << 20));
    for (int nt = 0; nt < 2; ++nt)
      for (int between = 0; between < 2; ++between)          // 64 MB of unrelated traffic between the passes
        for (int flip = 0; flip < 2; ++flip) {
          float best = 1e9, sum = 0;
          const int reps = 40;
          for (int r = 0; r < reps + 4; ++r) {
            if (between) pass_kernel<<<208, 512>>>((const floatx4*)other, (64l << 20) / 32768, 0, nt, sink);
            CK(hipEventRecord(e0));
            pass_kernel<<<208, 512>>>(buf, nb, flip ? (r & 1) : 0, nt, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 4) { sum += ms; best = ms < best ? ms : best; }
          }
          printf("%4ld MB  nt=%d  other-traffic=%d  %s: avg %.1f us (%.2f TB/s)  best %.1f us\n", mb, nt, between,
                 flip ? "flip" : "same", sum / reps * 1e3, mb * 1.048576e6 / (sum / reps * 1e-3) * 1e-12, best * 1e3);
        }
    CK(hipFree(buf)); CK(hipFree(other));
  }
  return 0;
}
