// LDS-DMA / wait-count / barrier helpers shared by the MFMA GEMM kernels (gfx950 only).
#pragma once
#include "common.hpp"

namespace mmssl {

// global -> LDS without a VGPR round trip: each lane moves 16 bytes from its own global address to
// LDS[m0 + 16 * lane] (the destination is lane-linear, so any swizzle goes on the SOURCE address).
// M0 is compiler-reserved and not preserved around inline asm: save / restore it.
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep_m0;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep_m0)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// same with the non-temporal hint (an operand streamed once should not push other kernels' tables out of L2)
__device__ __forceinline__ void glds16_nt(const float* gsrc, unsigned lds_dst) {
  unsigned keep_m0;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
      : "=&s"(keep_m0)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
// the same from a wave-uniform base (SGPR pair) + a 32-bit per-lane byte offset: no 64-bit per-lane pointer to carry
__device__ __forceinline__ void glds16_s(const char* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep_m0;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep_m0)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait_n() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// raw s_barrier: __syncthreads() would also drain vmcnt, i.e. the LDS-DMA pieces meant to stay in flight across it
__device__ __forceinline__ void bare_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// lgkmcnt(0) as the BUILTIN (simm16: vmcnt = 63, expcnt = 7, lgkmcnt = 0): the compiler's wait-count pass sees it,
// so it does not put redundant s_waitcnt instructions between the dependent MFMAs that follow
__device__ __forceinline__ void lgkm_wait0() {
  __builtin_amdgcn_s_waitcnt(0xC07F);
  asm volatile("" ::: "memory");
}

}  // namespace mmssl
