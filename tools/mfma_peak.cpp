// Pure-register fp32 MFMA throughput on gfx950 (no memory traffic): what can v_mfma_f32_32x32x2_f32 sustain?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.cpp -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
  floatx16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    x += 1e-9f;
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks, int iters, const char* tag) {
  float* out;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 16 * NACC * 2.0 * 32 * 32 * 2;
  printf("%-34s blocks=%5d  %.3f ms  %.1f TFLOP/s\n", tag, blocks, ms, flops / ms * 1e-9);
  hipFree(out);
}

int main() {
  run<1>(256 * 1, 4000, "1 acc, 1 block/CU (1 wave/SIMD)");
  run<1>(256 * 4, 2000, "1 acc, 4 blocks/CU (4 waves/SIMD)");
  run<4>(256 * 1, 2000, "4 acc, 1 block/CU");
  run<4>(256 * 3, 1000, "4 acc, 3 blocks/CU");
  run<1>(256 * 4, 40, "1 acc, 4 blocks/CU, short (100us)");
  return 0;
}
