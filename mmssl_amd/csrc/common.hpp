// Shared host/device helpers for libmmssl_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mmssl_hip.h"

#define MMSSL_HIP_TRY(expr)                       \
  do {                                            \
    hipError_t _e = (expr);                       \
    if (_e != hipSuccess) return (int)_e;         \
  } while (0)

// Kernel launches never sync; surface launch-config errors only.
#define MMSSL_LAUNCH_CHECK()                      \
  do {                                            \
    hipError_t _e = hipGetLastError();            \
    if (_e != hipSuccess) return (int)_e;         \
  } while (0)

namespace mmssl {

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 waves: one per SIMD of a CU

struct __attribute__((aligned(8))) Edge {
  int32_t col;
  float val;
};

// ---- lane-group (sub-wave) all-reduce over groups of W consecutive lanes, W in {8,16,32,64}.
// Butterfly with xor shuffles: every lane ends with the group's result.
template <int W>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int m = 1; m < W; m <<= 1) v += __shfl_xor(v, m, kWave);
  return v;
}
template <int W>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int m = 1; m < W; m <<= 1) v = fmaxf(v, __shfl_xor(v, m, kWave));
  return v;
}
// Sum across the 64/W groups of a wave, lane-position-wise (lane l with l^W, l^2W, ...).
template <int W>
__device__ __forceinline__ float cross_group_sum(float v) {
#pragma unroll
  for (int m = W; m < kWave; m <<= 1) v += __shfl_xor(v, m, kWave);
  return v;
}

__device__ __forceinline__ float4 f4_fma(float s, float4 x, float4 a) {
  a.x = fmaf(s, x.x, a.x);
  a.y = fmaf(s, x.y, a.y);
  a.z = fmaf(s, x.z, a.z);
  a.w = fmaf(s, x.w, a.w);
  return a;
}
__device__ __forceinline__ float f4_dot(float4 a, float4 b) {
  return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// softmax over the 4*LPR features of a row held by one lane group (the last GCN layer, Models.py:203-204): used by the
// SpMM's store epilogue and by the stand-alone row kernel, which therefore agree bit for bit
template <int LPR>
__device__ __forceinline__ float4 row_softmax(float4 a) {
  float m = fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w));
  m = group_max<LPR>(m);
  a.x = expf(a.x - m);
  a.y = expf(a.y - m);
  a.z = expf(a.z - m);
  a.w = expf(a.w - m);
  const float s = group_sum<LPR>((a.x + a.y) + (a.z + a.w));
  a.x /= s;
  a.y /= s;
  a.z /= s;
  a.w /= s;
  return a;
}

// Deterministic block-wide sum for kBlock threads; result valid in thread 0. `red` = 4 floats LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = group_sum<64>(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) r = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return r;
}

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

inline bool supported_d(int d) { return d == 32 || d == 64 || d == 128 || d == 256; }

// csrc/simtopk.hip: batch similarity rows with a CSR mask given as (rowptr, column array with `m_stride` bytes
// between consecutive columns)
int sim_launch(const float* Q, const int64_t* qidx, int64_t B, const float* T, int64_t I, int d, const int32_t* m_rowptr,
               const void* m_cols, int m_stride, float mask_value, const float* row_scale, float* out, int64_t ldo,
               float* sumsq_part, hipStream_t s);
// the row factors 1 / max(|masked score row|, eps) without forming the rows (Gram matrix of T; workspace bytes below)
size_t usim_norms_workspace(int d, int64_t n_items);
int usim_norms_launch(const float* Q, const int64_t* qidx, int64_t B, const float* T, int64_t I, int d,
                      const int32_t* m_rowptr, const void* m_cols, int m_stride, float eps, float* inv_out, void* ws,
                      hipStream_t s);

}  // namespace mmssl
