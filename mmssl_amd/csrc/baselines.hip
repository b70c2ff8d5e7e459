// Kernels of the baseline models the reference ships next to MMSSL (SURVEY.md 8f "next #4"): LATTICE / MICRO item-graph
// propagation over k-nearest-neighbour LISTS and the NGCF layer's elementwise tail.
//
//   item-graph product   /root/reference/LATTICE/codes/Models.py:103-104, MICRO/codes/Models.py:60-64, 112-118
//       h' = item_adj . h with item_adj a kNN graph: kept as lists idx [N, k] / w [N, k] (ELL) instead of the
//       reference's dense N x N (or COO) matrix.  forward = ELL SpMM, backward = its transpose by fp32 atomics for h and
//       an SDDMM (one dot product per stored entry) for the LEARNED weights w.
//   NGCF layer           LATTICE/codes/Models.py:106-118, MICRO/codes/Models.py:126-139, 195-204
//       bi_in = ego * side                                             (ngcf_mul)
//       ego'  = dropout(leaky_relu(GC(side)) + leaky_relu(Bi(bi_in)));  norm = normalize(ego')      (ngcf_combine)
//       the two nn.Linear products run on the projection kernels (csrc/linear.hip).
// HBM-bound row kernels: one lane group of d/4 lanes x float4 per row (d in {32, 64, 128, 256}).
#include "common.hpp"

using namespace mmssl;

namespace {

constexpr float kLeakySlope = 0.01f;        // F.leaky_relu's default negative slope

template <int LPR>
__global__ __launch_bounds__(kBlock) void ell_spmm_kernel(const int64_t* __restrict__ idx, const float* __restrict__ w,
                                                          int64_t rows, int k, const float4* __restrict__ H,
                                                          float4* __restrict__ Y) {
  constexpr int GPB = kBlock / LPR;
  const int lig = threadIdx.x & (LPR - 1);
  for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; row < rows; row += (int64_t)gridDim.x * GPB) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int j = 0; j < k; ++j) {                        // neighbours in list order: fixed summation order
      const int64_t c = idx[row * k + j];
      acc = f4_fma(w[row * k + j], H[c * LPR + lig], acc);
    }
    Y[row * LPR + lig] = acc;
  }
}

// gW[i, j] = < gY[i], H[idx[i, j]] >  (SDDMM on the stored pattern);  gH[idx[i, j]] += w[i, j] * gY[i]  (atomics)
template <int LPR>
__global__ __launch_bounds__(kBlock) void ell_spmm_bwd_kernel(const int64_t* __restrict__ idx, const float* __restrict__ w,
                                                              int64_t rows, int k, const float4* __restrict__ H,
                                                              const float4* __restrict__ gY, float* __restrict__ gW,
                                                              float* __restrict__ gH) {
  constexpr int GPB = kBlock / LPR;
  const int lig = threadIdx.x & (LPR - 1);
  for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; row < rows; row += (int64_t)gridDim.x * GPB) {
    const float4 g = gY[row * LPR + lig];
    for (int j = 0; j < k; ++j) {
      const int64_t c = idx[row * k + j];
      if (gW) {
        const float s = group_sum<LPR>(f4_dot(g, H[c * LPR + lig]));
        if (lig == 0) gW[row * k + j] = s;
      }
      if (gH) {
        const float ww = w[row * k + j];
        float* dst = gH + (c * LPR + lig) * 4;
        unsafeAtomicAdd(dst + 0, ww * g.x);
        unsafeAtomicAdd(dst + 1, ww * g.y);
        unsafeAtomicAdd(dst + 2, ww * g.z);
        unsafeAtomicAdd(dst + 3, ww * g.w);
      }
    }
  }
}

// out = a * b  /  ga = g * b, gb = g * a
__global__ __launch_bounds__(kBlock) void mul_kernel(const float4* __restrict__ a, const float4* __restrict__ b, int64_t n4,
                                                     float4* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 x = a[i], y = b[i];
    out[i] = make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w);
  }
}
__global__ __launch_bounds__(kBlock) void mul_bwd_kernel(const float4* __restrict__ a, const float4* __restrict__ b,
                                                         const float4* __restrict__ g, int64_t n4, float4* __restrict__ ga,
                                                         float4* __restrict__ gb) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 x = a[i], y = b[i], gg = g[i];
    ga[i] = make_float4(gg.x * y.x, gg.y * y.y, gg.z * y.z, gg.w * y.w);
    gb[i] = make_float4(gg.x * x.x, gg.y * x.y, gg.z * x.z, gg.w * x.w);
  }
}

__device__ __forceinline__ float lrelu(float x) { return x > 0.f ? x : kLeakySlope * x; }
__device__ __forceinline__ float lrelu_grad(float x) { return x > 0.f ? 1.f : kLeakySlope; }

// ego = keep ? (lrelu(G) + lrelu(B)) * scale : 0 ;  norm = ego / max(|ego|, eps)
template <int LPR>
__global__ __launch_bounds__(kBlock) void ngcf_combine_kernel(const float4* __restrict__ G, const float4* __restrict__ B,
                                                              const uchar4* __restrict__ keep, float scale, int64_t rows,
                                                              float eps, float4* __restrict__ ego,
                                                              float4* __restrict__ norm) {
  constexpr int GPB = kBlock / LPR;
  const int lig = threadIdx.x & (LPR - 1);
  for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; row < rows; row += (int64_t)gridDim.x * GPB) {
    const int64_t o = row * LPR + lig;
    const float4 g = G[o], b = B[o];
    float4 e = make_float4(lrelu(g.x) + lrelu(b.x), lrelu(g.y) + lrelu(b.y), lrelu(g.z) + lrelu(b.z),
                           lrelu(g.w) + lrelu(b.w));
    if (keep) {
      const uchar4 kk = keep[o];
      e = make_float4(kk.x ? e.x * scale : 0.f, kk.y ? e.y * scale : 0.f, kk.z ? e.z * scale : 0.f,
                      kk.w ? e.w * scale : 0.f);
    }
    const float inv = 1.f / fmaxf(sqrtf(group_sum<LPR>(f4_dot(e, e))), eps);
    ego[o] = e;
    norm[o] = make_float4(e.x * inv, e.y * inv, e.z * inv, e.w * inv);
  }
}

// gE = g_ego + normalize_bwd(ego, g_norm);  gG = gE * mask * lrelu'(G), gB = gE * mask * lrelu'(B)   (either g may be NULL)
template <int LPR>
__global__ __launch_bounds__(kBlock) void ngcf_combine_bwd_kernel(const float4* __restrict__ G, const float4* __restrict__ B,
                                                                  const uchar4* __restrict__ keep, float scale,
                                                                  const float4* __restrict__ ego,
                                                                  const float4* __restrict__ g_ego,
                                                                  const float4* __restrict__ g_norm, int64_t rows,
                                                                  float eps, float4* __restrict__ gG,
                                                                  float4* __restrict__ gB) {
  constexpr int GPB = kBlock / LPR;
  const int lig = threadIdx.x & (LPR - 1);
  for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; row < rows; row += (int64_t)gridDim.x * GPB) {
    const int64_t o = row * LPR + lig;
    const float4 e = ego[o];
    float4 ge = g_ego ? g_ego[o] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (g_norm) {
      const float4 gn = g_norm[o];
      const float ss = group_sum<LPR>(f4_dot(e, e));
      const float eg = group_sum<LPR>(f4_dot(e, gn));
      const float nrm = sqrtf(ss);
      float a, b;                     // normalize_bwd = a * gn - b * e
      if (nrm >= eps) {
        a = 1.f / nrm;
        b = eg / (nrm * ss);
      } else {
        a = 1.f / eps;
        b = 0.f;
      }
      ge = make_float4(ge.x + a * gn.x - b * e.x, ge.y + a * gn.y - b * e.y, ge.z + a * gn.z - b * e.z,
                       ge.w + a * gn.w - b * e.w);
    }
    if (keep) {
      const uchar4 kk = keep[o];
      ge = make_float4(kk.x ? ge.x * scale : 0.f, kk.y ? ge.y * scale : 0.f, kk.z ? ge.z * scale : 0.f,
                       kk.w ? ge.w * scale : 0.f);
    }
    const float4 g = G[o], b = B[o];
    gG[o] = make_float4(ge.x * lrelu_grad(g.x), ge.y * lrelu_grad(g.y), ge.z * lrelu_grad(g.z), ge.w * lrelu_grad(g.w));
    gB[o] = make_float4(ge.x * lrelu_grad(b.x), ge.y * lrelu_grad(b.y), ge.z * lrelu_grad(b.z), ge.w * lrelu_grad(b.w));
  }
}

inline unsigned row_grid(int64_t rows, int lpr) {
  const int64_t gpb = kBlock / lpr;
  int64_t nb = (rows + gpb - 1) / gpb;
  return (unsigned)(nb < 1 ? 1 : (nb > 8192 ? 8192 : nb));
}

#define ROW_DISPATCH(KERNEL, grid, s, ...)                                                      \
  switch (d) {                                                                                  \
    case 32: hipLaunchKernelGGL((KERNEL<8>), grid, dim3(kBlock), 0, s, __VA_ARGS__); break;     \
    case 64: hipLaunchKernelGGL((KERNEL<16>), grid, dim3(kBlock), 0, s, __VA_ARGS__); break;    \
    case 128: hipLaunchKernelGGL((KERNEL<32>), grid, dim3(kBlock), 0, s, __VA_ARGS__); break;   \
    case 256: hipLaunchKernelGGL((KERNEL<64>), grid, dim3(kBlock), 0, s, __VA_ARGS__); break;   \
    default: return MMSSL_E_UNSUPP;                                                             \
  }

}  // namespace

extern "C" int mmssl_ell_spmm_f32(const int64_t* idx, const float* w, int64_t rows, int k, const float* H, int d, float* Y,
                                  void* stream) {
  if (rows < 0 || k < 1 || (rows > 0 && (!idx || !w || !H || !Y))) return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  if (rows == 0) return 0;
  if (((uintptr_t)H | (uintptr_t)Y) & 15) return MMSSL_E_BADARG;
  hipStream_t s = as_stream(stream);
  const dim3 grid(row_grid(rows, d / 4));
  ROW_DISPATCH(ell_spmm_kernel, grid, s, idx, w, rows, k, reinterpret_cast<const float4*>(H), reinterpret_cast<float4*>(Y));
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_ell_spmm_bwd_f32(const int64_t* idx, const float* w, int64_t rows, int k, const float* H, int d,
                                      const float* gY, float* gW, float* gH, void* stream) {
  if (rows < 0 || k < 1 || (rows > 0 && (!idx || !w || !H || !gY)) || (!gW && !gH)) return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  if (rows == 0) return 0;
  if (((uintptr_t)H | (uintptr_t)gY | (uintptr_t)gH) & 15) return MMSSL_E_BADARG;
  hipStream_t s = as_stream(stream);
  const dim3 grid(row_grid(rows, d / 4));
  ROW_DISPATCH(ell_spmm_bwd_kernel, grid, s, idx, w, rows, k, reinterpret_cast<const float4*>(H),
               reinterpret_cast<const float4*>(gY), gW, gH);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_mul_f32(const float* a, const float* b, int64_t n, float* out, void* stream) {
  if (n < 0 || (n & 3) || (n > 0 && (!a || !b || !out))) return MMSSL_E_BADARG;
  if (n == 0) return 0;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) return MMSSL_E_BADARG;
  int64_t nb = (n / 4 + kBlock - 1) / kBlock;
  nb = nb > 8192 ? 8192 : nb;
  hipLaunchKernelGGL(mul_kernel, dim3((unsigned)nb), dim3(kBlock), 0, as_stream(stream), reinterpret_cast<const float4*>(a),
                     reinterpret_cast<const float4*>(b), n / 4, reinterpret_cast<float4*>(out));
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_mul_bwd_f32(const float* a, const float* b, const float* g, int64_t n, float* ga, float* gb,
                                 void* stream) {
  if (n < 0 || (n & 3) || (n > 0 && (!a || !b || !g || !ga || !gb))) return MMSSL_E_BADARG;
  if (n == 0) return 0;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)g | (uintptr_t)ga | (uintptr_t)gb) & 15) return MMSSL_E_BADARG;
  int64_t nb = (n / 4 + kBlock - 1) / kBlock;
  nb = nb > 8192 ? 8192 : nb;
  hipLaunchKernelGGL(mul_bwd_kernel, dim3((unsigned)nb), dim3(kBlock), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(a), reinterpret_cast<const float4*>(b),
                     reinterpret_cast<const float4*>(g), n / 4, reinterpret_cast<float4*>(ga), reinterpret_cast<float4*>(gb));
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_ngcf_combine_f32(const float* G, const float* B, const uint8_t* keep, float scale, int64_t rows, int d,
                                      float eps, float* ego, float* norm, void* stream) {
  if (rows < 0 || !(eps > 0.f) || (rows > 0 && (!G || !B || !ego || !norm))) return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  if (rows == 0) return 0;
  if ((((uintptr_t)G | (uintptr_t)B | (uintptr_t)ego | (uintptr_t)norm) & 15) || ((uintptr_t)keep & 3)) return MMSSL_E_BADARG;
  hipStream_t s = as_stream(stream);
  const dim3 grid(row_grid(rows, d / 4));
  ROW_DISPATCH(ngcf_combine_kernel, grid, s, reinterpret_cast<const float4*>(G), reinterpret_cast<const float4*>(B),
               reinterpret_cast<const uchar4*>(keep), scale, rows, eps, reinterpret_cast<float4*>(ego),
               reinterpret_cast<float4*>(norm));
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_ngcf_combine_bwd_f32(const float* G, const float* B, const uint8_t* keep, float scale, const float* ego,
                                          const float* g_ego, const float* g_norm, int64_t rows, int d, float eps,
                                          float* gG, float* gB, void* stream) {
  if (rows < 0 || !(eps > 0.f) || (rows > 0 && (!G || !B || !ego || !gG || !gB)) || (!g_ego && !g_norm)) return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  if (rows == 0) return 0;
  if ((((uintptr_t)G | (uintptr_t)B | (uintptr_t)ego | (uintptr_t)g_ego | (uintptr_t)g_norm | (uintptr_t)gG | (uintptr_t)gB) & 15) ||
      ((uintptr_t)keep & 3))
    return MMSSL_E_BADARG;
  hipStream_t s = as_stream(stream);
  const dim3 grid(row_grid(rows, d / 4));
  ROW_DISPATCH(ngcf_combine_bwd_kernel, grid, s, reinterpret_cast<const float4*>(G), reinterpret_cast<const float4*>(B),
               reinterpret_cast<const uchar4*>(keep), scale, reinterpret_cast<const float4*>(ego),
               reinterpret_cast<const float4*>(g_ego), reinterpret_cast<const float4*>(g_norm), rows, eps,
               reinterpret_cast<float4*>(gG), reinterpret_cast<float4*>(gB));
  MMSSL_LAUNCH_CHECK();
  return 0;
}
