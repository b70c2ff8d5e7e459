// MFMA fed from LDS exactly like gemm64_kernel's inner loop (no global traffic): where does the time go?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
constexpr int LD = 65, BK = 32;

template <int MODE>   // 0: reads+mfma+barrier per slice  1: no barrier  2: reads only once (registers)  3: 2 acc per wave
__global__ __launch_bounds__(256) void k(float* out, int slices) {
  __shared__ float As[2][BK * LD];
  __shared__ float Bs[2][BK * LD];
  for (int i = threadIdx.x; i < 2 * BK * LD; i += 256) { (&As[0][0])[i] = i * 1e-4f; (&Bs[0][0])[i] = 1.f + i * 1e-5f; }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int frag = (lane >> 5) * LD + (lane & 31);
  floatx16 acc, acc2;
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
  float fa[16], fb[16];
  for (int s = 0; s < 16; ++s) { fa[s] = As[0][frag + wm * 32 + 2 * s * LD]; fb[s] = Bs[0][frag + wn * 32 + 2 * s * LD]; }
  for (int kt = 0; kt < slices; ++kt) {
    const int buf = kt & 1;
    const float* as = As[buf] + frag + wm * 32;
    const float* bs = Bs[buf] + frag + wn * 32;
    if (MODE != 2) {
#pragma unroll
      for (int s = 0; s < 16; ++s) { fa[s] = as[2 * s * LD]; fb[s] = bs[2 * s * LD]; }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[s], acc, 0, 0, 0);
      if (MODE == 3) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[s], fa[s], acc2, 0, 0, 0);
    }
    if (MODE == 0 || MODE == 3) __syncthreads();
  }
  float sum = 0.f;
  for (int r = 0; r < 16; ++r) sum += acc[r] + acc2[r];
  out[blockIdx.x * 256 + threadIdx.x] = sum;
}

template <int MODE>
void run(int blocks, int slices, const char* tag) {
  float* out; hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, slices); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, slices); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 4 * slices * 16 * (MODE == 3 ? 2 : 1) * 2.0 * 32 * 32 * 2;
  printf("%-52s %.3f ms %.1f TF\n", tag, ms, fl / ms * 1e-9);
  hipFree(out);
}
int main() {
  run<0>(1024, 400, "LDS reads + 16 MFMA + barrier per slice, 4 blk/CU");
  run<1>(1024, 400, "same, no barrier");
  run<2>(1024, 400, "no LDS reads in loop (regs), no barrier");
  run<3>(1024, 400, "2 accumulators (32 MFMA per slice) + barrier");
  run<0>(768, 400, "mode0 with 3 blk/CU");
  run<0>(512, 400, "mode0 with 2 blk/CU");
  run<0>(256, 400, "mode0 with 1 blk/CU");
  return 0;
}
