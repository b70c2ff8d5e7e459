// Row-wise kernels: L2 normalise (+fused scale/add), its backward, softmax backward, sum of
// squares. All HBM-bound streaming: one lane group (d/4 lanes, one float4 each) per row, so
// every row access is a single coalesced request; reductions are wavefront shuffles.
//
// Reference call sites (/root/reference/MMSSL/): F.normalize at Models.py:196-197,217-218 and
// main.py:212-213; softmax at Models.py:203-204; (x**2).sum() at main.py:252-257,503.
#include "common.hpp"

using namespace mmssl;

namespace {

// Y = alpha * X / max(||X||, eps) (+ Base)
template <int LPR>
__global__ __launch_bounds__(kBlock) void l2norm_fwd_kernel(const float4* __restrict__ X,
                                                            const float4* __restrict__ Base,
                                                            float alpha, int64_t rows, float eps,
                                                            float4* __restrict__ Y) {
  constexpr int GPB = kBlock / LPR;
  const int lig = threadIdx.x & (LPR - 1);
  const int64_t stride = (int64_t)gridDim.x * GPB;
  for (int64_t r = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; r < rows; r += stride) {
    const float4 x = X[r * LPR + lig];
    const float ss = group_sum<LPR>(f4_dot(x, x));
    const float s = alpha / fmaxf(sqrtf(ss), eps);
    float4 y = make_float4(x.x * s, x.y * s, x.z * s, x.w * s);
    if (Base) {
      const float4 b = Base[r * LPR + lig];
      y.x += b.x; y.y += b.y; y.z += b.z; y.w += b.w;
    }
    Y[r * LPR + lig] = y;
  }
}

// y = alpha * x / den, den = max(norm, eps):
//   norm >= eps : gx = alpha * (g / norm - x * (x.g) / norm^3)
//   norm <  eps : gx = alpha * g / eps            (clamp_min passes no gradient to norm)
template <int LPR>
__global__ __launch_bounds__(kBlock) void l2norm_bwd_kernel(const float4* __restrict__ X,
                                                            const float4* __restrict__ G,
                                                            float alpha, int64_t rows, float eps,
                                                            float4* __restrict__ GX) {
  constexpr int GPB = kBlock / LPR;
  const int lig = threadIdx.x & (LPR - 1);
  const int64_t stride = (int64_t)gridDim.x * GPB;
  for (int64_t r = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; r < rows; r += stride) {
    const float4 x = X[r * LPR + lig];
    const float4 g = G[r * LPR + lig];
    const float ss = group_sum<LPR>(f4_dot(x, x));
    const float xg = group_sum<LPR>(f4_dot(x, g));
    const float norm = sqrtf(ss);
    float a, b;  // gx = a*g - b*x
    if (norm >= eps) {
      a = alpha / norm;
      b = alpha * xg / (norm * ss);
    } else {
      a = alpha / eps;
      b = 0.f;
    }
    GX[r * LPR + lig] = make_float4(a * g.x - b * x.x, a * g.y - b * x.y, a * g.z - b * x.z,
                                    a * g.w - b * x.w);
  }
}

// Y = softmax(X) row by row (in place allowed): the SpMM epilogue's arithmetic as a launch of its own, for products whose
// rows only become complete after a reduce-scatter / after all column chunks have arrived (mmssl_amd/dist.py)
template <int LPR>
__global__ __launch_bounds__(kBlock) void softmax_fwd_kernel(const float4* X, int64_t rows, float4* Y) {
  constexpr int GPB = kBlock / LPR;
  const int lig = threadIdx.x & (LPR - 1);
  const int64_t stride = (int64_t)gridDim.x * GPB;
  for (int64_t r = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; r < rows; r += stride)
    Y[r * LPR + lig] = row_softmax<LPR>(X[r * LPR + lig]);
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void softmax_bwd_kernel(const float4* __restrict__ Yv,
                                                             const float4* __restrict__ G, float scale,
                                                             int64_t rows, float4* __restrict__ GX) {
  constexpr int GPB = kBlock / LPR;
  const int lig = threadIdx.x & (LPR - 1);
  const int64_t stride = (int64_t)gridDim.x * GPB;
  for (int64_t r = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; r < rows; r += stride) {
    const float4 y = Yv[r * LPR + lig];
    const float4 g = G[r * LPR + lig];
    const float s = group_sum<LPR>(f4_dot(y, g));
    GX[r * LPR + lig] = make_float4(scale * y.x * (g.x - s), scale * y.y * (g.y - s), scale * y.z * (g.z - s),
                                    scale * y.w * (g.w - s));
  }
}

// ---- layer mean + modality fusion (+ feature-regulariser partial sums) ----------------------
constexpr int kMaxLayers = 8;
struct LayerPtrs {
  const float4* p[kMaxLayers];
};

// out = inv * sum_k L_k + r * A/max(|A|,eps) + r * B/max(|B|,eps);  part[block] = sum(|A|^2+|B|^2)
template <int LPR>
__global__ __launch_bounds__(kBlock) void combine_fwd_kernel(LayerPtrs L, int n_layers, float inv,
                                                             const float4* __restrict__ A,
                                                             const float4* __restrict__ B, float r, int64_t rows,
                                                             float eps, float4* __restrict__ out,
                                                             float* __restrict__ part) {
  __shared__ float red[4];
  constexpr int GPB = kBlock / LPR;
  const int lig = threadIdx.x & (LPR - 1);
  const int64_t stride = (int64_t)gridDim.x * GPB;
  float ss = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; row < rows; row += stride) {
    const int64_t o = row * LPR + lig;
    float4 acc = L.p[0][o];
    for (int k = 1; k < n_layers; ++k) {
      const float4 v = L.p[k][o];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float4 a = A[o], b = B[o];
    const float sa = group_sum<LPR>(f4_dot(a, a));
    const float sb = group_sum<LPR>(f4_dot(b, b));
    const float ca = r / fmaxf(sqrtf(sa), eps), cb = r / fmaxf(sqrtf(sb), eps);
    out[o] = make_float4(fmaf(inv, acc.x, fmaf(ca, a.x, cb * b.x)), fmaf(inv, acc.y, fmaf(ca, a.y, cb * b.y)),
                         fmaf(inv, acc.z, fmaf(ca, a.z, cb * b.z)), fmaf(inv, acc.w, fmaf(ca, a.w, cb * b.w)));
    if (lig == 0) ss += sa + sb;
  }
  if (part) {                       // uniform branch: every thread of the block takes it
    const float t = block_sum_256(ss, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
  }
}

// gA = r * normalize_bwd(A, G) + c * A ; gB likewise ; gL = inv * G
template <int LPR>
__global__ __launch_bounds__(kBlock) void combine_bwd_kernel(const float4* __restrict__ A,
                                                             const float4* __restrict__ B,
                                                             const float4* __restrict__ G, float r, float inv,
                                                             const float* __restrict__ c_dev, float c_scale,
                                                             int64_t rows, float eps, float4* __restrict__ gA,
                                                             float4* __restrict__ gB, float4* __restrict__ gL) {
  constexpr int GPB = kBlock / LPR;
  const int lig = threadIdx.x & (LPR - 1);
  const int64_t stride = (int64_t)gridDim.x * GPB;
  const float c = c_dev ? c_scale * c_dev[0] : 0.f;
  for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR; row < rows; row += stride) {
    const int64_t o = row * LPR + lig;
    const float4 g = G[o];
    if (gL) gL[o] = make_float4(inv * g.x, inv * g.y, inv * g.z, inv * g.w);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const float4 x = which ? B[o] : A[o];
      const float ss = group_sum<LPR>(f4_dot(x, x));
      const float xg = group_sum<LPR>(f4_dot(x, g));
      const float norm = sqrtf(ss);
      float a, b;                     // r*normalize_bwd = a*g - b*x
      if (norm >= eps) {
        a = r / norm;
        b = r * xg / (norm * ss);
      } else {
        a = r / eps;
        b = 0.f;
      }
      const float4 y = make_float4(a * g.x - (b - c) * x.x, a * g.y - (b - c) * x.y, a * g.z - (b - c) * x.z,
                                   a * g.w - (b - c) * x.w);
      (which ? gB : gA)[o] = y;
    }
  }
}

// ---- the same over PACKED modal features: Mod [rows, NM * d], NM modalities side by side --------------------------
// One lane group of LPR * NM lanes per row: sub-group m (LPR lanes) owns the row's slice of modality m (its norm is a
// sub-group reduction), the modalities meet by xor-shuffles across sub-groups; sub-group m also adds the layers
// m, m + NM, ... so that every lane streams. Up to two SIDES (user tables, item tables) share one launch:
// blocks [0, S0.blocks) run side 0, the rest side 1.
struct FuseSide {
  LayerPtrs L;
  const float4* Mod;
  const float4* G;      // backward only
  const float4* Gx;     // backward only, may be NULL
  float4* out;          // forward: fused rows; backward: gMod
  float4* gL;           // backward only, may be NULL
  float* part;          // may be NULL: per-block sums of |Mod|^2 (both directions see every row of Mod)
  const int64_t* idx;   // forward only, may be NULL: the rows to compute (then `rows` = their number)
  int64_t lo, n_local;  // with idx: entries are GLOBAL row ids of a row-sharded table; this rank holds rows [lo, lo + n_local)
  //                       as local rows 0.. and skips the others (lo = 0, n_local = max: plain row lists)
  int64_t rows;
  int n_layers;
  int blocks;
};

template <int LPR, int NM>
__global__ __launch_bounds__(kBlock) void fuse_fwd_kernel(FuseSide S0, FuseSide S1, float inv, float r, float eps) {
  __shared__ float red[4];
  constexpr int GL = LPR * NM;                 // lanes per row
  constexpr int GPB = kBlock / GL;
  const bool second = (int)blockIdx.x >= S0.blocks;
  const FuseSide& S = second ? S1 : S0;
  const int blk = second ? (int)blockIdx.x - S0.blocks : (int)blockIdx.x;
  const int lig = threadIdx.x & (LPR - 1);
  const int m = (threadIdx.x & (GL - 1)) / LPR;
  const int64_t stride = (int64_t)S.blocks * GPB;
  float ss = 0.f;
  for (int64_t it = (int64_t)blk * GPB + threadIdx.x / GL; it < S.rows; it += stride) {
    const int64_t row = S.idx ? S.idx[it] - S.lo : it;
    if (S.idx && (row < 0 || row >= S.n_local)) continue;      // a row another rank owns (uniform over the row's lanes)
    const int64_t o = row * LPR + lig;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = m; k < S.n_layers; k += NM) {
      const float4 v = S.L.p[k][o];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const float4 a = S.Mod[(row * NM + m) * LPR + lig];
    const float sa = group_sum<LPR>(f4_dot(a, a));
    const float ca = r / fmaxf(sqrtf(sa), eps);
    float4 y = make_float4(fmaf(inv, acc.x, ca * a.x), fmaf(inv, acc.y, ca * a.y), fmaf(inv, acc.z, ca * a.z),
                           fmaf(inv, acc.w, ca * a.w));
    float srow = sa;
#pragma unroll
    for (int w = LPR; w < GL; w <<= 1) {       // sub-groups meet: butterfly over the modality index (fixed order)
      y.x += __shfl_xor(y.x, w, kWave); y.y += __shfl_xor(y.y, w, kWave);
      y.z += __shfl_xor(y.z, w, kWave); y.w += __shfl_xor(y.w, w, kWave);
      srow += __shfl_xor(srow, w, kWave);
    }
    if (m == 0) S.out[o] = y;
    if ((threadIdx.x & (GL - 1)) == 0) ss += srow;
  }
  if (S.part) {                     // block-uniform branch
    const float t = block_sum_256(ss, red);
    if (threadIdx.x == 0) S.part[blk] = t;
  }
}

template <int LPR, int NM>
__global__ __launch_bounds__(kBlock) void fuse_bwd_kernel(FuseSide S0, FuseSide S1, float r, float inv,
                                                          const float* __restrict__ c_dev, float c_scale, float eps) {
  constexpr int GL = LPR * NM;
  constexpr int GPB = kBlock / GL;
  const bool second = (int)blockIdx.x >= S0.blocks;
  const FuseSide& S = second ? S1 : S0;
  const int blk = second ? (int)blockIdx.x - S0.blocks : (int)blockIdx.x;
  const int lig = threadIdx.x & (LPR - 1);
  const int m = (threadIdx.x & (GL - 1)) / LPR;
  const int64_t stride = (int64_t)S.blocks * GPB;
  const float c = c_dev ? c_scale * c_dev[0] : 0.f;
  __shared__ float red[4];
  float ssacc = 0.f;
  for (int64_t row = (int64_t)blk * GPB + threadIdx.x / GL; row < S.rows; row += stride) {
    const int64_t o = row * LPR + lig;
    const int64_t mo = (row * NM + m) * LPR + lig;
    const float4 g = S.G[o];
    const float4 x = S.Mod[mo];
    if (S.gL && m == 0) S.gL[o] = make_float4(inv * g.x, inv * g.y, inv * g.z, inv * g.w);
    const float ss = group_sum<LPR>(f4_dot(x, x));
    const float xg = group_sum<LPR>(f4_dot(x, g));
    if (lig == 0) ssacc += ss;
    const float norm = sqrtf(ss);
    float a, b;                     // r*normalize_bwd = a*g - b*x
    if (norm >= eps) {
      a = r / norm;
      b = r * xg / (norm * ss);
    } else {
      a = r / eps;
      b = 0.f;
    }
    float4 y = make_float4(a * g.x - (b - c) * x.x, a * g.y - (b - c) * x.y, a * g.z - (b - c) * x.z,
                           a * g.w - (b - c) * x.w);
    if (S.Gx) {
      const float4 e = S.Gx[mo];
      y.x += e.x; y.y += e.y; y.z += e.z; y.w += e.w;
    }
    S.out[mo] = y;
  }
  if (S.part) {                     // block-uniform branch: the regulariser's |Mod|^2 sums fall out of the norms
    const float t = block_sum_256(ssacc, red);
    if (threadIdx.x == 0) S.part[blk] = t;
  }
}

__global__ __launch_bounds__(kBlock) void sum_partials_kernel(const float* __restrict__ part, int64_t n,
                                                              float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kBlock) acc += part[i];
  const float t = block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = t;
}

constexpr int kSumsqBlocks = 1024;

__global__ __launch_bounds__(kBlock) void sumsq_stage1(const float* __restrict__ X, int64_t n,
                                                       float* __restrict__ part) {
  __shared__ float red[4];
  const int64_t n4 = n >> 2;
  const float4* X4 = reinterpret_cast<const float4*>(X);
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 v = X4[i];
    acc += f4_dot(v, v);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const float v = X[(n4 << 2) + threadIdx.x];
    acc += v * v;
  }
  const float t = block_sum_256(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ __launch_bounds__(kBlock) void sumsq_stage2(const float* __restrict__ part, int nparts,
                                                       float* __restrict__ out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += kBlock) acc += part[i];
  const float t = block_sum_256(acc, red);
  if (threadIdx.x == 0) out[0] = t;
}

inline int row_grid(int64_t rows, int lpr) {
  const int64_t gpb = kBlock / lpr;
  const int64_t nb = (rows + gpb - 1) / gpb;
  return (int)(nb < 1 ? 1 : (nb > 256 * 16 ? 256 * 16 : nb));
}

}  // namespace

#define ROW_DISPATCH(KERNEL, ...)                                                                 \
  switch (d) {                                                                                    \
    case 32: hipLaunchKernelGGL((KERNEL<8>), dim3(row_grid(rows, 8)), dim3(kBlock), 0, s, __VA_ARGS__); break;    \
    case 64: hipLaunchKernelGGL((KERNEL<16>), dim3(row_grid(rows, 16)), dim3(kBlock), 0, s, __VA_ARGS__); break;  \
    case 128: hipLaunchKernelGGL((KERNEL<32>), dim3(row_grid(rows, 32)), dim3(kBlock), 0, s, __VA_ARGS__); break; \
    case 256: hipLaunchKernelGGL((KERNEL<64>), dim3(row_grid(rows, 64)), dim3(kBlock), 0, s, __VA_ARGS__); break; \
    default: return MMSSL_E_UNSUPP;                                                               \
  }

extern "C" int mmssl_l2norm_rows_f32(const float* X, const float* Base, float alpha, int64_t rows, int d,
                                     float eps, float* Y, void* stream) {
  if (rows < 0 || (rows > 0 && (!X || !Y))) return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  if (rows == 0) return 0;
  hipStream_t s = as_stream(stream);
  ROW_DISPATCH(l2norm_fwd_kernel, reinterpret_cast<const float4*>(X), reinterpret_cast<const float4*>(Base),
               alpha, rows, eps, reinterpret_cast<float4*>(Y));
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_l2norm_rows_bwd_f32(const float* X, const float* gY, float alpha, int64_t rows, int d,
                                         float eps, float* gX, void* stream) {
  if (rows < 0 || (rows > 0 && (!X || !gY || !gX))) return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  if (rows == 0) return 0;
  hipStream_t s = as_stream(stream);
  ROW_DISPATCH(l2norm_bwd_kernel, reinterpret_cast<const float4*>(X), reinterpret_cast<const float4*>(gY),
               alpha, rows, eps, reinterpret_cast<float4*>(gX));
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_softmax_rows_f32(const float* X, int64_t rows, int d, float* Y, void* stream) {
  if (rows < 0 || (rows > 0 && (!X || !Y))) return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  if (((uintptr_t)X | (uintptr_t)Y) & 15) return MMSSL_E_BADARG;
  if (rows == 0) return 0;
  hipStream_t s = as_stream(stream);
  ROW_DISPATCH(softmax_fwd_kernel, reinterpret_cast<const float4*>(X), rows, reinterpret_cast<float4*>(Y));
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_softmax_rows_bwd_f32(const float* Y, const float* gY, float scale, int64_t rows, int d,
                                          float* gX, void* stream) {
  if (rows < 0 || (rows > 0 && (!Y || !gY || !gX))) return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  if (rows == 0) return 0;
  hipStream_t s = as_stream(stream);
  ROW_DISPATCH(softmax_bwd_kernel, reinterpret_cast<const float4*>(Y), reinterpret_cast<const float4*>(gY),
               scale, rows, reinterpret_cast<float4*>(gX));
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t mmssl_sumsq_workspace_bytes(int64_t n) {
  (void)n;
  return (size_t)kSumsqBlocks * sizeof(float);
}

extern "C" int mmssl_sumsq_f32(const float* X, int64_t n, float* out, void* workspace, size_t workspace_bytes,
                               void* stream) {
  if (n < 0 || !out || (n > 0 && !X)) return MMSSL_E_BADARG;
  if (((uintptr_t)X) & 15) return MMSSL_E_BADARG;
  if (!workspace || workspace_bytes < mmssl_sumsq_workspace_bytes(n)) return MMSSL_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  int64_t nb = (n / 4 + kBlock - 1) / kBlock;
  nb = nb < 1 ? 1 : (nb > kSumsqBlocks ? kSumsqBlocks : nb);
  float* part = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(sumsq_stage1, dim3((int)nb), dim3(kBlock), 0, s, X, n, part);
  MMSSL_LAUNCH_CHECK();
  hipLaunchKernelGGL(sumsq_stage2, dim3(1), dim3(kBlock), 0, s, part, (int)nb, out);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_layer_combine_blocks(int64_t rows, int d) {
  if (!supported_d(d) || rows <= 0) return 0;
  return row_grid(rows, d / 4);
}

extern "C" int mmssl_layer_combine_f32(const float* const* layers, int n_layers, float inv, const float* A,
                                       const float* B, float r, int64_t rows, int d, float eps, float* out,
                                       float* sumsq_part, void* stream) {
  if (rows < 0 || n_layers < 1 || n_layers > kMaxLayers || !layers || (rows > 0 && (!A || !B || !out)))
    return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  if (rows == 0) return 0;
  LayerPtrs L;
  for (int k = 0; k < kMaxLayers; ++k)
    L.p[k] = reinterpret_cast<const float4*>(k < n_layers ? layers[k] : layers[0]);
  for (int k = 0; k < n_layers; ++k)
    if (!layers[k]) return MMSSL_E_BADARG;
  hipStream_t s = as_stream(stream);
  ROW_DISPATCH(combine_fwd_kernel, L, n_layers, inv, reinterpret_cast<const float4*>(A),
               reinterpret_cast<const float4*>(B), r, rows, eps, reinterpret_cast<float4*>(out), sumsq_part);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

namespace {
inline int fuse_grid(int64_t rows, int lanes_per_row) {
  const int64_t gpb = kBlock / lanes_per_row;
  const int64_t nb = (rows + gpb - 1) / gpb;
  return (int)(nb < 1 ? 1 : (nb > 256 * 16 ? 256 * 16 : nb));
}
inline bool fuse_shape_ok(int d, int nm) {
  return supported_d(d) && (nm == 1 || nm == 2 || nm == 4) && (d / 4) * nm <= 64;
}

#define FUSE_DISPATCH(KERNEL, ...)                                                                            \
  switch ((d / 4) * 8 + nm) {                                                                                 \
    case 8 * 8 + 1: hipLaunchKernelGGL((KERNEL<8, 1>), grid, dim3(kBlock), 0, s, __VA_ARGS__); break;         \
    case 8 * 8 + 2: hipLaunchKernelGGL((KERNEL<8, 2>), grid, dim3(kBlock), 0, s, __VA_ARGS__); break;         \
    case 8 * 8 + 4: hipLaunchKernelGGL((KERNEL<8, 4>), grid, dim3(kBlock), 0, s, __VA_ARGS__); break;         \
    case 16 * 8 + 1: hipLaunchKernelGGL((KERNEL<16, 1>), grid, dim3(kBlock), 0, s, __VA_ARGS__); break;       \
    case 16 * 8 + 2: hipLaunchKernelGGL((KERNEL<16, 2>), grid, dim3(kBlock), 0, s, __VA_ARGS__); break;       \
    case 16 * 8 + 4: hipLaunchKernelGGL((KERNEL<16, 4>), grid, dim3(kBlock), 0, s, __VA_ARGS__); break;       \
    case 32 * 8 + 1: hipLaunchKernelGGL((KERNEL<32, 1>), grid, dim3(kBlock), 0, s, __VA_ARGS__); break;       \
    case 32 * 8 + 2: hipLaunchKernelGGL((KERNEL<32, 2>), grid, dim3(kBlock), 0, s, __VA_ARGS__); break;       \
    case 64 * 8 + 1: hipLaunchKernelGGL((KERNEL<64, 1>), grid, dim3(kBlock), 0, s, __VA_ARGS__); break;       \
    default: return MMSSL_E_UNSUPP;                                                                           \
  }
}  // namespace

// sides: 1 or 2 (user tables, item tables) in one launch
extern "C" int mmssl_fuse_fwd_f32(int sides, const float* const* const* layers, int n_layers, float inv,
                                  const float* const* Mod, int nm, float r, const int64_t* rows, int d, float eps,
                                  float* const* out, float* const* sumsq_part, void* stream) {
  if (sides < 1 || sides > 2 || n_layers < 1 || n_layers > kMaxLayers || !layers || !Mod || !rows || !out)
    return MMSSL_E_BADARG;
  if (!fuse_shape_ok(d, nm)) return MMSSL_E_UNSUPP;
  FuseSide S[2] = {};
  int total = 0;
  for (int k = 0; k < sides; ++k) {
    if (rows[k] <= 0 || !layers[k] || !Mod[k] || !out[k]) return MMSSL_E_BADARG;
    if (((uintptr_t)Mod[k] | (uintptr_t)out[k]) & 15) return MMSSL_E_BADARG;
    for (int l = 0; l < n_layers; ++l)
      if (!layers[k][l]) return MMSSL_E_BADARG;
    for (int l = 0; l < kMaxLayers; ++l)
      S[k].L.p[l] = reinterpret_cast<const float4*>(layers[k][l < n_layers ? l : 0]);
    S[k].Mod = reinterpret_cast<const float4*>(Mod[k]);
    S[k].out = reinterpret_cast<float4*>(out[k]);
    S[k].part = sumsq_part ? sumsq_part[k] : nullptr;
    S[k].rows = rows[k];
    S[k].n_layers = n_layers;
    S[k].blocks = fuse_grid(rows[k], (d / 4) * nm);
    total += S[k].blocks;
  }
  hipStream_t s = as_stream(stream);
  const dim3 grid((unsigned)total);
  FUSE_DISPATCH(fuse_fwd_kernel, S[0], S[1], inv, r, eps);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

// number of per-block partial sums side k of mmssl_fuse_fwd_f32 writes
extern "C" int mmssl_fuse_blocks(int64_t rows, int d, int nm) {
  if (!fuse_shape_ok(d, nm) || rows <= 0) return 0;
  return fuse_grid(rows, (d / 4) * nm);
}

// The same rows arithmetic for a LIST of rows per side (idx[k]: n_idx[k] int64 row numbers, repeats allowed): only those
// rows of out[k] are written, bit for bit what mmssl_fuse_fwd_f32 writes there. A training step needs the fused tables
// at its batch rows only; the dense launch can then leave the step's critical path (the |Mod|^2 sums of the regulariser
// come out of mmssl_fuse_bwd_f32).
extern "C" int mmssl_fuse_fwd_rows_f32(int sides, const float* const* const* layers, int n_layers, float inv,
                                       const float* const* Mod, int nm, float r, const int64_t* const* idx,
                                       const int64_t* n_idx, int d, float eps, float* const* out, void* stream) {
  return mmssl_fuse_fwd_owned_rows_f32(sides, layers, n_layers, inv, Mod, nm, r, idx, n_idx, nullptr, nullptr, d, eps, out,
                                       stream);
}

extern "C" int mmssl_fuse_fwd_owned_rows_f32(int sides, const float* const* const* layers, int n_layers, float inv,
                                             const float* const* Mod, int nm, float r, const int64_t* const* idx,
                                             const int64_t* n_idx, const int64_t* lo, const int64_t* n_local, int d,
                                             float eps, float* const* out, void* stream) {
  if (sides < 1 || sides > 2 || n_layers < 1 || n_layers > kMaxLayers || !layers || !Mod || !idx || !n_idx || !out)
    return MMSSL_E_BADARG;
  if ((lo == nullptr) != (n_local == nullptr)) return MMSSL_E_BADARG;
  if (!fuse_shape_ok(d, nm)) return MMSSL_E_UNSUPP;
  FuseSide S[2] = {};
  int total = 0;
  for (int k = 0; k < sides; ++k) {
    if (n_idx[k] <= 0 || !layers[k] || !Mod[k] || !out[k] || !idx[k]) return MMSSL_E_BADARG;
    if (((uintptr_t)Mod[k] | (uintptr_t)out[k]) & 15) return MMSSL_E_BADARG;
    for (int l = 0; l < n_layers; ++l)
      if (!layers[k][l]) return MMSSL_E_BADARG;
    for (int l = 0; l < kMaxLayers; ++l)
      S[k].L.p[l] = reinterpret_cast<const float4*>(layers[k][l < n_layers ? l : 0]);
    S[k].Mod = reinterpret_cast<const float4*>(Mod[k]);
    S[k].out = reinterpret_cast<float4*>(out[k]);
    S[k].part = nullptr;
    S[k].idx = idx[k];
    S[k].lo = lo ? lo[k] : 0;
    S[k].n_local = n_local ? n_local[k] : INT64_MAX;
    if (S[k].n_local < 0) return MMSSL_E_BADARG;
    S[k].rows = n_idx[k];
    S[k].n_layers = n_layers;
    S[k].blocks = fuse_grid(n_idx[k], (d / 4) * nm);
    total += S[k].blocks;
  }
  hipStream_t s = as_stream(stream);
  const dim3 grid((unsigned)total);
  FUSE_DISPATCH(fuse_fwd_kernel, S[0], S[1], inv, r, eps);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

namespace {
// total[0] += c * sum(part), sum_out[0] = sum(part): the regulariser joins a loss that was assembled without it
__global__ __launch_bounds__(kBlock) void loss_add_partials_kernel(const float* __restrict__ part, int64_t n, float c,
                                                                   float* __restrict__ total,
                                                                   float* __restrict__ sum_out) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += kBlock) acc += part[i];
  const float t = block_sum_256(acc, red);
  if (threadIdx.x == 0) {
    total[0] += c * t;
    if (sum_out) sum_out[0] = t;
  }
}
}  // namespace

extern "C" int mmssl_loss_add_partials_f32(const float* part, int64_t n, float c, float* total, float* sum_out,
                                           void* stream) {
  if (!part || n < 1 || !total) return MMSSL_E_BADARG;
  hipLaunchKernelGGL(loss_add_partials_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), part, n, c, total, sum_out);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_fuse_bwd_f32(int sides, const float* const* Mod, int nm, const float* const* G,
                                  const float* const* Gx, float r, float inv, const float* c_dev, float c_scale,
                                  const int64_t* rows, int d, float eps, float* const* gMod, float* const* gL,
                                  float* const* sumsq_part, void* stream) {
  if (sides < 1 || sides > 2 || !Mod || !G || !rows || !gMod) return MMSSL_E_BADARG;
  if (!fuse_shape_ok(d, nm)) return MMSSL_E_UNSUPP;
  FuseSide S[2] = {};
  int total = 0;
  for (int k = 0; k < sides; ++k) {
    if (rows[k] <= 0 || !Mod[k] || !G[k] || !gMod[k]) return MMSSL_E_BADARG;
    const float* gx = Gx ? Gx[k] : nullptr;
    float* gl = gL ? gL[k] : nullptr;
    if (((uintptr_t)Mod[k] | (uintptr_t)G[k] | (uintptr_t)gx | (uintptr_t)gMod[k] | (uintptr_t)gl) & 15) return MMSSL_E_BADARG;
    S[k].Mod = reinterpret_cast<const float4*>(Mod[k]);
    S[k].G = reinterpret_cast<const float4*>(G[k]);
    S[k].Gx = reinterpret_cast<const float4*>(gx);
    S[k].out = reinterpret_cast<float4*>(gMod[k]);
    S[k].gL = reinterpret_cast<float4*>(gl);
    S[k].part = sumsq_part ? sumsq_part[k] : nullptr;
    S[k].rows = rows[k];
    S[k].blocks = fuse_grid(rows[k], (d / 4) * nm);
    total += S[k].blocks;
  }
  hipStream_t s = as_stream(stream);
  const dim3 grid((unsigned)total);
  FUSE_DISPATCH(fuse_bwd_kernel, S[0], S[1], r, inv, c_dev, c_scale, eps);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

namespace {
struct CombineSide {
  const float4* A;
  const float4* B;
  const float4* G;
  float4* gA;
  float4* gB;
  float4* gL;
  int64_t rows;
  int blocks;
};

// both sides (user tables, item tables) of the fusion backward in ONE launch: blocks [0, S0.blocks) run side 0, the rest
// side 1, each with the arithmetic (and the grid-stride order) of combine_bwd_kernel on its own grid
template <int LPR>
__global__ __launch_bounds__(kBlock) void combine_bwd2_kernel(CombineSide S0, CombineSide S1, float r, float inv,
                                                              const float* __restrict__ c_dev, float c_scale, float eps) {
  constexpr int GPB = kBlock / LPR;
  const bool second = (int)blockIdx.x >= S0.blocks;
  const CombineSide& S = second ? S1 : S0;
  const int blk = second ? (int)blockIdx.x - S0.blocks : (int)blockIdx.x;
  const int lig = threadIdx.x & (LPR - 1);
  const int64_t stride = (int64_t)S.blocks * GPB;
  const float c = c_dev ? c_scale * c_dev[0] : 0.f;
  for (int64_t row = (int64_t)blk * GPB + threadIdx.x / LPR; row < S.rows; row += stride) {
    const int64_t o = row * LPR + lig;
    const float4 g = S.G[o];
    if (S.gL) S.gL[o] = make_float4(inv * g.x, inv * g.y, inv * g.z, inv * g.w);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      const float4 x = which ? S.B[o] : S.A[o];
      const float ss = group_sum<LPR>(f4_dot(x, x));
      const float xg = group_sum<LPR>(f4_dot(x, g));
      const float norm = sqrtf(ss);
      float a, b;                     // r*normalize_bwd = a*g - b*x
      if (norm >= eps) {
        a = r / norm;
        b = r * xg / (norm * ss);
      } else {
        a = r / eps;
        b = 0.f;
      }
      const float4 y = make_float4(a * g.x - (b - c) * x.x, a * g.y - (b - c) * x.y, a * g.z - (b - c) * x.z,
                                   a * g.w - (b - c) * x.w);
      (which ? S.gB : S.gA)[o] = y;
    }
  }
}
}  // namespace

extern "C" int mmssl_layer_combine_bwd2_f32(const float* A0, const float* B0, const float* G0, int64_t rows0, float* gA0,
                                            float* gB0, float* gL0, const float* A1, const float* B1, const float* G1,
                                            int64_t rows1, float* gA1, float* gB1, float* gL1, float r, float inv,
                                            const float* c_dev, float c_scale, int d, float eps, void* stream) {
  if (rows0 <= 0 || rows1 <= 0 || !A0 || !B0 || !G0 || !gA0 || !gB0 || !A1 || !B1 || !G1 || !gA1 || !gB1)
    return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  const int lpr = d / 4;
  CombineSide S0{reinterpret_cast<const float4*>(A0), reinterpret_cast<const float4*>(B0),
                 reinterpret_cast<const float4*>(G0), reinterpret_cast<float4*>(gA0), reinterpret_cast<float4*>(gB0),
                 reinterpret_cast<float4*>(gL0), rows0, (int)row_grid(rows0, lpr)};
  CombineSide S1{reinterpret_cast<const float4*>(A1), reinterpret_cast<const float4*>(B1),
                 reinterpret_cast<const float4*>(G1), reinterpret_cast<float4*>(gA1), reinterpret_cast<float4*>(gB1),
                 reinterpret_cast<float4*>(gL1), rows1, (int)row_grid(rows1, lpr)};
  hipStream_t s = as_stream(stream);
  const dim3 grid((unsigned)(S0.blocks + S1.blocks));
  switch (d) {
    case 32: hipLaunchKernelGGL((combine_bwd2_kernel<8>), grid, dim3(kBlock), 0, s, S0, S1, r, inv, c_dev, c_scale, eps); break;
    case 64: hipLaunchKernelGGL((combine_bwd2_kernel<16>), grid, dim3(kBlock), 0, s, S0, S1, r, inv, c_dev, c_scale, eps); break;
    case 128: hipLaunchKernelGGL((combine_bwd2_kernel<32>), grid, dim3(kBlock), 0, s, S0, S1, r, inv, c_dev, c_scale, eps); break;
    case 256: hipLaunchKernelGGL((combine_bwd2_kernel<64>), grid, dim3(kBlock), 0, s, S0, S1, r, inv, c_dev, c_scale, eps); break;
  }
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_layer_combine_bwd_f32(const float* A, const float* B, const float* G, float r, float inv,
                                           const float* c_dev, float c_scale, int64_t rows, int d, float eps,
                                           float* gA, float* gB, float* gL, void* stream) {
  if (rows < 0 || (rows > 0 && (!A || !B || !G || !gA || !gB))) return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  if (rows == 0) return 0;
  hipStream_t s = as_stream(stream);
  ROW_DISPATCH(combine_bwd_kernel, reinterpret_cast<const float4*>(A), reinterpret_cast<const float4*>(B),
               reinterpret_cast<const float4*>(G), r, inv, c_dev, c_scale, rows, eps,
               reinterpret_cast<float4*>(gA), reinterpret_cast<float4*>(gB), reinterpret_cast<float4*>(gL));
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_sum_partials_f32(const float* part, int64_t n, float* out, void* stream) {
  if (n < 0 || !out || (n > 0 && !part)) return MMSSL_E_BADARG;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(kBlock), 0, as_stream(stream), part, n, out);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

namespace {
__global__ void loss_assemble_kernel(const float* __restrict__ terms, const float* __restrict__ w, int n,
                                     const float* __restrict__ extra, float c, float* __restrict__ total) {
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < n; ++k) t += w[k] * terms[k];
    if (extra) t += c * extra[0];
    total[0] = t;
  }
}
}  // namespace

namespace {
// loss assembly + the step's counter ticks in ONE launch: this kernel sits between the forward (whose dropout
// launch read the RNG launch counter) and the optimiser (whose bias correction reads the step counters), so it can
// advance all of them without a launch of their own (3 one-thread kernels per step otherwise).
struct TickPtrs {
  float* f32[4];
  unsigned long long* u64[4];
  int n_f32, n_u64;
};
__global__ void loss_assemble_tick_kernel(const float* __restrict__ terms, const float* __restrict__ w, int n,
                                          const float* __restrict__ extra, float c, float* __restrict__ total,
                                          TickPtrs T) {
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < n; ++k) t += w[k] * terms[k];
    if (extra) t += c * extra[0];
    total[0] = t;
    for (int k = 0; k < T.n_f32; ++k) T.f32[k][0] += 1.0f;
    for (int k = 0; k < T.n_u64; ++k) T.u64[k][0] += 1ull;
  }
}
}  // namespace

extern "C" int mmssl_loss_assemble_tick_f32(const float* terms, const float* w, int n, const float* extra, float c,
                                            float* total, float* const* f32_ticks, int n_f32,
                                            uint64_t* const* u64_ticks, int n_u64, void* stream) {
  if (!terms || !w || !total || n < 0 || n > 16 || n_f32 < 0 || n_f32 > 4 || n_u64 < 0 || n_u64 > 4) return MMSSL_E_BADARG;
  if ((n_f32 > 0 && !f32_ticks) || (n_u64 > 0 && !u64_ticks)) return MMSSL_E_BADARG;
  TickPtrs T;
  T.n_f32 = n_f32;
  T.n_u64 = n_u64;
  for (int k = 0; k < 4; ++k) {
    T.f32[k] = k < n_f32 ? f32_ticks[k] : nullptr;
    T.u64[k] = k < n_u64 ? reinterpret_cast<unsigned long long*>(u64_ticks[k]) : nullptr;
    if ((k < n_f32 && !T.f32[k]) || (k < n_u64 && !T.u64[k])) return MMSSL_E_BADARG;
  }
  hipLaunchKernelGGL(loss_assemble_tick_kernel, dim3(1), dim3(64), 0, as_stream(stream), terms, w, n, extra, c, total, T);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

namespace {
__global__ void loss_assemble_bwd_kernel(const float* __restrict__ g, const float* __restrict__ w, int n, float c,
                                         float* __restrict__ gterms, float* __restrict__ gextra) {
  const float gv = g[0];
  if ((int)threadIdx.x < n) gterms[threadIdx.x] = gv * w[threadIdx.x];
  if (threadIdx.x == 0 && gextra) gextra[0] = gv * c;
}
}  // namespace

extern "C" int mmssl_loss_assemble_bwd_f32(const float* g, const float* w, int n, float c, float* gterms,
                                           float* gextra, void* stream) {
  if (!g || !w || !gterms || n < 0 || n > 16) return MMSSL_E_BADARG;
  hipLaunchKernelGGL(loss_assemble_bwd_kernel, dim3(1), dim3(64), 0, as_stream(stream), g, w, n, c, gterms, gextra);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_loss_assemble_f32(const float* terms, const float* w, int n, const float* extra, float c,
                                       float* total, void* stream) {
  if (!terms || !w || !total || n < 0 || n > 16) return MMSSL_E_BADARG;
  hipLaunchKernelGGL(loss_assemble_kernel, dim3(1), dim3(64), 0, as_stream(stream), terms, w, n, extra, c, total);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

namespace {
// out = g * keep * scale (dropout backward), 4 elements per thread
__global__ __launch_bounds__(kBlock) void mask_scale_kernel(const float4* __restrict__ g,
                                                            const uchar4* __restrict__ keep, float scale,
                                                            int64_t n4, float4* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 v = g[i];
    const uchar4 k = keep[i];
    out[i] = make_float4(k.x ? v.x * scale : 0.f, k.y ? v.y * scale : 0.f, k.z ? v.z * scale : 0.f,
                         k.w ? v.w * scale : 0.f);
  }
}
}  // namespace

extern "C" int mmssl_mask_scale_f32(const float* g, const uint8_t* keep, float scale, int64_t n, float* out,
                                    void* stream) {
  if (n < 0 || (n & 3) || (n > 0 && (!g || !keep || !out))) return MMSSL_E_BADARG;
  if (n == 0) return 0;
  int64_t nb = (n / 4 + kBlock - 1) / kBlock;
  nb = nb > 4096 ? 4096 : nb;
  hipLaunchKernelGGL(mask_scale_kernel, dim3((unsigned)nb), dim3(kBlock), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(g), reinterpret_cast<const uchar4*>(keep), scale, n / 4,
                     reinterpret_cast<float4*>(out));
  MMSSL_LAUNCH_CHECK();
  return 0;
}

// Dropout backward of PACKED modal rows: G [rows, nm * dm] holds nm modalities side by side, keep is the uint8
// [nm, rows, dm] mask of mmssl_proj_fwd_f32. out may alias G. (The unsharded step gets this from the SpMM's mask
// epilogue; the row-sharded step applies it after the reduce-scatter of the partial products.)
namespace {
__global__ __launch_bounds__(kBlock) void mask_packed_kernel(const float4* __restrict__ G, const uchar4* __restrict__ keep,
                                                             float scale, int64_t rows, int nm, int dm4,
                                                             float4* __restrict__ out) {
  const int64_t n4 = rows * nm * dm4;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const int64_t row = i / (nm * dm4);
    const int c = (int)(i - row * (nm * dm4));
    const int m = c / dm4, cc = c - m * dm4;
    const uchar4 k = keep[((int64_t)m * rows + row) * dm4 + cc];
    const float4 g = G[i];
    out[i] = make_float4(k.x ? g.x * scale : 0.f, k.y ? g.y * scale : 0.f, k.z ? g.z * scale : 0.f,
                         k.w ? g.w * scale : 0.f);
  }
}
}  // namespace

extern "C" int mmssl_mask_packed_f32(const float* G, const uint8_t* keep, float scale, int64_t rows, int nm, int dm,
                                     float* out, void* stream) {
  if (rows < 0 || nm < 1 || dm < 4 || (dm & 3) || (rows > 0 && (!G || !keep || !out))) return MMSSL_E_BADARG;
  if ((((uintptr_t)G | (uintptr_t)out) & 15) || ((uintptr_t)keep & 3)) return MMSSL_E_BADARG;
  if (rows == 0) return 0;
  const int64_t n4 = rows * nm * (dm / 4);
  int64_t nb = (n4 + kBlock - 1) / kBlock;
  nb = nb > 4096 ? 4096 : nb;
  hipLaunchKernelGGL(mask_packed_kernel, dim3((unsigned)nb), dim3(kBlock), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(G), reinterpret_cast<const uchar4*>(keep), scale, rows, nm, dm / 4,
                     reinterpret_cast<float4*>(out));
  MMSSL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// Rows of a ROW-SHARDED table for a batch of global indices (mmssl_amd/dist.py: the batch rows of BPR / InfoNCE
// are assembled from the row owners by an all-reduce of zero-padded buffers): out[j] = table[idx[j] - lo] when this
// rank owns row idx[j] (lo <= idx[j] < lo + rows_local), else zeros. The backward scatter-adds the gradient rows this
// rank owns into its (pre-zeroed) table gradient with hardware fp32 atomics (a batch may name a row twice).
// ---------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(kBlock) void gather_owned_rows_kernel(const float4* __restrict__ table, int64_t rows_local,
                                                                   int d4, const int64_t* __restrict__ idx, int64_t n,
                                                                   int64_t lo, float4* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n * d4) return;
  const int64_t j = i / d4;
  const int c = (int)(i - j * d4);
  const int64_t r = idx[j] - lo;
  out[i] = (r >= 0 && r < rows_local) ? table[r * d4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ __launch_bounds__(kBlock) void scatter_owned_rows_kernel(const float* __restrict__ g, int d,
                                                                    const int64_t* __restrict__ idx, int64_t n,
                                                                    int64_t lo, int64_t rows_local,
                                                                    float* __restrict__ gtable) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n * d) return;
  const int64_t j = i / d;
  const int c = (int)(i - j * d);
  const int64_t r = idx[j] - lo;
  if (r >= 0 && r < rows_local) atomicAdd(gtable + r * d + c, g[i]);
}
}  // namespace

extern "C" int mmssl_gather_owned_rows_f32(const float* table, int64_t rows_local, int d, const int64_t* idx, int64_t n,
                                           int64_t lo, float* out, void* stream) {
  if (rows_local < 0 || n < 0 || d <= 0 || (d & 3)) return MMSSL_E_BADARG;
  if (n == 0) return 0;
  if (!table || !idx || !out || (((uintptr_t)table | (uintptr_t)out) & 15)) return MMSSL_E_BADARG;
  const int64_t total = n * (d / 4);
  hipLaunchKernelGGL(gather_owned_rows_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     as_stream(stream), reinterpret_cast<const float4*>(table), rows_local, d / 4, idx, n, lo,
                     reinterpret_cast<float4*>(out));
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_scatter_owned_rows_f32(const float* g, const int64_t* idx, int64_t n, int64_t lo, int64_t rows_local,
                                            int d, float* gtable, void* stream) {
  if (rows_local < 0 || n < 0 || d <= 0) return MMSSL_E_BADARG;
  if (n == 0) return 0;
  if (!g || !idx || !gtable) return MMSSL_E_BADARG;
  const int64_t total = n * d;
  hipLaunchKernelGGL(scatter_owned_rows_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     as_stream(stream), g, d, idx, n, lo, rows_local, gtable);
  MMSSL_LAUNCH_CHECK();
  return 0;
}
