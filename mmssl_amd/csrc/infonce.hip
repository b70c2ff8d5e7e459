// Fused InfoNCE (cross-modal contrastive) loss, forward + backward, for gfx950.
// Replaces Trainer.sim + Trainer.batched_contrastive_loss
// (/root/reference/MMSSL/main.py:211-249): ~12 small PyTorch launches per call and two
// materialised n x n matrices become 3 (fwd) / 2 (bwd) launches with nothing n x n in memory.
//
//   n1 = z1/max(|z1|,eps), n2 = z2/max(|z2|,eps)
//   r_ij = exp(n1_i.n1_j / tau), b_ij = exp(n1_i.n2_j / tau)
//   D_i  = sum_{j!=i} r_ij + sum_j b_ij          (the reference adds r_ii then subtracts it)
//   loss = mean_i -log(b_ii / D_i + 1e-8)
//
// Tiling: a 256-thread block owns a 32-row tile and walks 32-column tiles staged in LDS
// (rows padded by 4 floats: ds_read_b128 conflict-free for d % 8 == 0). Dot tiles are plain
// fp32 VALU FMAs (2x2 micro-tiles); the row log-sum-exp terms are reduced across the 16 lanes
// that share a row with wavefront shuffles. Column ranges are split over blocks so that even
// n = 1024 fills the chip; split partials are summed in a fixed order (deterministic).
//
// Backward (w_i = -(1/n) q_i/(q_i+1e-8), c_i = w_i/(D_i tau)):
//   gn1_t = (w_t/tau) n2_t - sum_s c_t b_ts n2_s - sum_{s!=t} (c_t + c_s) r_ts n1_s
//   gn2_t = (w_t/tau) n1_t - sum_s c_s b_st n1_s
// followed by the F.normalize backward (incl. its g/eps branch for all-zero rows).
#include "common.hpp"
#include "bpr_step.hpp"
#include "lds_dma.hpp"

using namespace mmssl;

namespace {

constexpr int T = 32;        // tile edge (rows and columns)
constexpr float kNormEps = 1e-12f;
constexpr int CP = T + 4;    // padded row length of a coefficient tile

__host__ __device__ inline int n_tiles(int64_t n) { return (int)((n + T - 1) / T); }

// number of column splits: aim for >= ~512 blocks, at most one split per column tile
inline int splits_for(int64_t n, int col_tiles) {
  const int nt = n_tiles(n);
  int cs = (512 + nt - 1) / nt;
  if (cs > col_tiles) cs = col_tiles;
  if (cs < 1) cs = 1;
  return cs;
}

// MFMA path: column tiles per block in the forward (4 waves x 2; 4 x 1 measured slower: 16 vs 13 us, and twice the
// partials for finalize_rows) / pair tiles per block in the backward (4 x 1)
constexpr int kFwdTilesPerBlock = 8;
constexpr int kBwdTilesPerBlock = 4;

struct Layout {   // offsets in floats into the workspace
  size_t n1, n2, inv1, inv2, pos, w, c, rows_part, loss, meta, g1p, g2p, total;
  int cs_f, cs_b;
};

inline bool use_mfma(int d);

inline Layout make_layout(int64_t n, int d) {
  Layout L;
  const int nt = n_tiles(n);
  if (use_mfma(d)) {                // block-level splits: 8 column tiles / 4 pair tiles per block
    L.cs_f = (2 * nt + kFwdTilesPerBlock - 1) / kFwdTilesPerBlock;
    L.cs_b = (nt + kBwdTilesPerBlock - 1) / kBwdTilesPerBlock;
  } else {
    L.cs_f = splits_for(n, 2 * nt);   // forward walks 2*nt column tiles (n1 then n2)
    L.cs_b = splits_for(n, nt);
  }
  size_t o = 0;
  auto take = [&](size_t cnt) { size_t r = o; o += (cnt + 3) & ~(size_t)3; return r; };
  // n1 / n2 / c are padded to whole 32-row tiles (prep writes zero rows): the LDS-staged tile kernels move 32 x d images
  L.n1 = take((size_t)nt * T * d);
  L.n2 = take((size_t)nt * T * d);
  L.inv1 = take(n); L.inv2 = take(n); L.pos = take(n); L.w = take(n); L.c = take((size_t)nt * T);
  L.rows_part = take((size_t)L.cs_f * n);
  L.loss = take((size_t)nt + 4);      // loss partials: per 256-row block (finalize_rows) or per row tile (deferred row terms)
  L.meta = take(4);                   // [0] the constant inside the logarithm, [1] != 0: row terms deferred to the backward
  L.g1p = take((size_t)L.cs_b * n * d);
  L.g2p = take((size_t)L.cs_b * n * d);
  L.total = o;
  return L;
}

// ---- prep: normalise both inputs, positive-pair cosine --------------------------------
constexpr int kMaxProblems = 4;
struct Z1Ptrs {          // per-problem first operand (forward) / its gradient (backward)
  const float* z1[kMaxProblems];
  float* gz1[kMaxProblems];
};

// blockIdx.y = problem: problems share z2 and differ in z1 (the reference calls the loss once per
// modality with the same user embeddings, main.py:411-412); each has its own workspace slice.
// Guest blocks (bpr.n_blocks > 0, blockIdx.x >= prep_blocks): the ROWS part of the hot step's BPR tail (bpr_step.hpp) -
// it depends on nothing this launch computes, and this short launch has three quarters of the chip free.
__global__ __launch_bounds__(kBlock) void prep_kernel(Z1Ptrs Z, const float* __restrict__ z2,
                                                      const int64_t* __restrict__ idx, int64_t n, int d,
                                                      float* __restrict__ ws, size_t ws_stride, Layout L,
                                                      int prep_blocks, float log_eps, int defer_rows, BprStepArgs bpr) {
  if ((int)blockIdx.x >= prep_blocks) {
    if (blockIdx.y == 0) {
      if (d == 64) bpr_rows_block<16>(bpr, (int)blockIdx.x - prep_blocks);
      else bpr_rows_block<8>(bpr, (int)blockIdx.x - prep_blocks);
    }
    return;
  }
  const float* __restrict__ z1 = Z.z1[blockIdx.y];
  float* __restrict__ wsp = ws + (size_t)blockIdx.y * ws_stride;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    wsp[L.meta] = log_eps;
    wsp[L.meta + 1] = defer_rows ? 1.f : 0.f;
  }
  float* __restrict__ n1 = wsp + L.n1;
  float* __restrict__ n2 = wsp + L.n2;
  float* __restrict__ inv1 = wsp + L.inv1;
  float* __restrict__ inv2 = wsp + L.inv2;
  float* __restrict__ pos = wsp + L.pos;
  // one 16-lane group per row, lanes stride over the d/4 float4 chunks
  const int lig = threadIdx.x & 15;
  const int64_t r = (int64_t)blockIdx.x * (kBlock / 16) + threadIdx.x / 16;
  const int nch = d >> 2;
  if (r >= n) {                                  // pad rows of the last tile: zeros (the tile kernels read whole tiles)
    if (r < (int64_t)n_tiles(n) * T)
      for (int k = lig; k < nch; k += 16) {
        reinterpret_cast<float4*>(n1 + r * d)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(n2 + r * d)[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    return;
  }
  const int64_t src = idx ? idx[r] : r;          // fused gather: row r of the batch = table row idx[r]
  const float4* a = reinterpret_cast<const float4*>(z1 + src * d);
  const float4* b = reinterpret_cast<const float4*>(z2 + src * d);
  float s1 = 0.f, s2 = 0.f, s12 = 0.f;
  for (int k = lig; k < nch; k += 16) {
    const float4 x = a[k], y = b[k];
    s1 += f4_dot(x, x);
    s2 += f4_dot(y, y);
    s12 += f4_dot(x, y);
  }
  s1 = group_sum<16>(s1);
  s2 = group_sum<16>(s2);
  s12 = group_sum<16>(s12);
  const float i1 = 1.f / fmaxf(sqrtf(s1), kNormEps);
  const float i2 = 1.f / fmaxf(sqrtf(s2), kNormEps);
  float4* o1 = reinterpret_cast<float4*>(n1 + r * d);
  float4* o2 = reinterpret_cast<float4*>(n2 + r * d);
  float p = 0.f;
  for (int k = lig; k < nch; k += 16) {
    float4 x = a[k], y = b[k];
    x.x *= i1; x.y *= i1; x.z *= i1; x.w *= i1;
    y.x *= i2; y.y *= i2; y.z *= i2; y.w *= i2;
    o1[k] = x;
    o2[k] = y;
    p += f4_dot(x, y);
  }
  p = group_sum<16>(p);
  if (lig == 0) {
    inv1[r] = i1;
    inv2[r] = i2;
    pos[r] = p;
  }
}

// stage a T x D tile (rows row0.., zero-filled past n) into LDS with row stride D+4
template <int D>
__device__ __forceinline__ void stage_tile(float* __restrict__ lds, const float* __restrict__ src,
                                           int64_t row0, int64_t n) {
  constexpr int S = D + 4;
  constexpr int CH = D / 4;                 // float4 chunks per row
  for (int idx = threadIdx.x; idx < T * CH; idx += kBlock) {
    const int r = idx / CH, k = idx - r * CH;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < n) v = reinterpret_cast<const float4*>(src + (row0 + r) * D)[k];
    *reinterpret_cast<float4*>(lds + r * S + k * 4) = v;
  }
}

// 2x2 micro-tile of dot products: rows {ty, ty+16} of A x rows {tx, tx+16} of B
template <int D>
__device__ __forceinline__ void dot_2x2(const float* __restrict__ A, const float* __restrict__ B, int ty,
                                        int tx, float (&o)[4]) {
  constexpr int S = D + 4;
  o[0] = o[1] = o[2] = o[3] = 0.f;
#pragma unroll 4
  for (int k = 0; k < D; k += 4) {
    const float4 a0 = *reinterpret_cast<const float4*>(A + ty * S + k);
    const float4 a1 = *reinterpret_cast<const float4*>(A + (ty + 16) * S + k);
    const float4 b0 = *reinterpret_cast<const float4*>(B + tx * S + k);
    const float4 b1 = *reinterpret_cast<const float4*>(B + (tx + 16) * S + k);
    o[0] += f4_dot(a0, b0);
    o[1] += f4_dot(a0, b1);
    o[2] += f4_dot(a1, b0);
    o[3] += f4_dot(a1, b1);
  }
}

// ---- forward: partial denominators per (row, column split) ------------------------------
template <int D>
__global__ __launch_bounds__(kBlock) void fwd_tiles_kernel(const float* __restrict__ ws, size_t ws_stride,
                                                           Layout L, int64_t n, float tau, int cs) {
  const float* __restrict__ wsp = ws + (size_t)blockIdx.y * ws_stride;
  const float* __restrict__ n1 = wsp + L.n1;
  const float* __restrict__ n2 = wsp + L.n2;
  float* __restrict__ rows_part = const_cast<float*>(wsp) + L.rows_part;
  constexpr int S = D + 4;
  __shared__ __attribute__((aligned(16))) float A[T * S];
  __shared__ __attribute__((aligned(16))) float B[T * S];
  const int nt = n_tiles(n);
  const int ti = blockIdx.x % nt;
  const int split = blockIdx.x / nt;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t i0 = (int64_t)ti * T;
  stage_tile<D>(A, n1, i0, n);
  // column tiles 0..nt-1 come from n1 (reflexive), nt..2nt-1 from n2 (between)
  const int per = (2 * nt + cs - 1) / cs;
  const int c_beg = split * per, c_end = min(2 * nt, c_beg + per);
  float acc0 = 0.f, acc1 = 0.f;     // rows ty and ty+16
  const int64_t gi0 = i0 + ty, gi1 = i0 + ty + 16;
  for (int ct = c_beg; ct < c_end; ++ct) {
    const bool refl = ct < nt;
    const int64_t j0 = (int64_t)(refl ? ct : ct - nt) * T;
    __syncthreads();                 // A staged / previous B fully consumed
    stage_tile<D>(B, refl ? n1 : n2, j0, n);
    __syncthreads();
    float dts[4];
    dot_2x2<D>(A, B, ty, tx, dts);
    const int64_t gj0 = j0 + tx, gj1 = j0 + tx + 16;
    const float e00 = (gj0 < n && !(refl && gj0 == gi0)) ? expf(dts[0] / tau) : 0.f;
    const float e01 = (gj1 < n && !(refl && gj1 == gi0)) ? expf(dts[1] / tau) : 0.f;
    const float e10 = (gj0 < n && !(refl && gj0 == gi1)) ? expf(dts[2] / tau) : 0.f;
    const float e11 = (gj1 < n && !(refl && gj1 == gi1)) ? expf(dts[3] / tau) : 0.f;
    acc0 += e00 + e01;
    acc1 += e10 + e11;
  }
  acc0 = group_sum<16>(acc0);        // the 16 lanes (tx) that share row ty
  acc1 = group_sum<16>(acc1);
  if (tx == 0) {
    if (gi0 < n) rows_part[(size_t)split * n + gi0] = acc0;
    if (gi1 < n) rows_part[(size_t)split * n + gi1] = acc1;
  }
}

// Row terms of row i from the forward's partial denominators: loss term li, w_i = dL/dlog-ratio / n, c_i = w_i / (D_i tau)
// (returned). One arithmetic for finalize_rows_kernel and for the backward tiles that compute it on the fly.
__device__ __forceinline__ float row_terms(const float* __restrict__ rows_part, const float* __restrict__ pos, int cs,
                                           int64_t n, int64_t i, float tau, float log_eps, float& wi, float& li) {
  const float ps = pos[i];
  float Dn = 0.f;
  int s = 0;
  for (; s + 8 <= cs; s += 8) {          // eight partials per L2 round trip; the adds stay in the order s = 0, 1, ...
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = rows_part[(size_t)(s + e) * n + i];
#pragma unroll
    for (int e = 0; e < 8; ++e) Dn += v[e];
  }
  for (; s < cs; ++s) Dn += rows_part[(size_t)s * n + i];
  const float q = expf(ps / tau) / Dn;
  li = -logf(q + log_eps);
  wi = -(q / (q + log_eps)) / (float)n;
  return wi / (Dn * tau);
}

// ---- finalize: per-row backward coefficients (row-parallel) + loss (fixed-order two-stage sum) ----
// `tickets` (one int per problem, 0 on entry, left 0) != NULL: the last block of a problem to arrive also reduces the
// row terms to the loss (the arithmetic of finalize_loss_kernel), which then is not launched.
__global__ __launch_bounds__(kBlock) void finalize_rows_kernel(float* __restrict__ ws, size_t ws_stride, Layout L,
                                                               int cs, int64_t n, float tau, float log_eps,
                                                               int* __restrict__ tickets, float* __restrict__ losses) {
  float* __restrict__ wsp = ws + (size_t)blockIdx.y * ws_stride;
  const float* __restrict__ rows_part = wsp + L.rows_part;
  const float* __restrict__ pos = wsp + L.pos;
  float* __restrict__ w = wsp + L.w;
  float* __restrict__ c = wsp + L.c;
  float* __restrict__ loss_part = wsp + L.loss;
  __shared__ float red[4];
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  float li = 0.f;
  if (i < n) {
    float wi;
    c[i] = row_terms(rows_part, pos, cs, n, i, tau, log_eps, wi, li);
    w[i] = wi;
  }
  const float t = block_sum_256(li, red);
  if (tickets == nullptr) {
    if (threadIdx.x == 0) loss_part[blockIdx.x] = t;
    return;
  }
  __shared__ int s_last;
  if (threadIdx.x == 0) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(loss_part) + blockIdx.x, __float_as_uint(t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    const int prev = __hip_atomic_fetch_add(tickets + blockIdx.y, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (prev == (int)gridDim.x - 1);
    if (s_last) __hip_atomic_store(tickets + blockIdx.y, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  float acc = 0.f;
  for (int k = threadIdx.x; k < (int)gridDim.x; k += kBlock)
    acc += __uint_as_float(__hip_atomic_load(reinterpret_cast<unsigned*>(loss_part) + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  const float tt = block_sum_256(acc, red);
  if (threadIdx.x == 0) losses[blockIdx.y] = tt / (float)n;
}
__global__ __launch_bounds__(kBlock) void finalize_loss_kernel(const float* __restrict__ ws, size_t ws_stride,
                                                               Layout L, int nparts, int64_t n,
                                                               float* __restrict__ loss) {
  const float* __restrict__ loss_part = ws + (size_t)blockIdx.y * ws_stride + L.loss;
  __shared__ float red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += kBlock) acc += loss_part[i];
  const float t = block_sum_256(acc, red);
  if (threadIdx.x == 0) loss[blockIdx.y] = t / (float)n;
}

// ---- backward tiles --------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(kBlock) void bwd_tiles_kernel(float* __restrict__ ws, size_t ws_stride, Layout L,
                                                           int64_t n, float tau, int cs) {
  float* __restrict__ wsp = ws + (size_t)blockIdx.y * ws_stride;
  const float* __restrict__ n1 = wsp + L.n1;
  const float* __restrict__ n2 = wsp + L.n2;
  const float* __restrict__ c = wsp + L.c;
  float* __restrict__ g1p = wsp + L.g1p;
  float* __restrict__ g2p = wsp + L.g2p;
  constexpr int S = D + 4;
  constexpr int FPT = D / 16;        // output features per thread in the accumulate phase
  __shared__ __attribute__((aligned(16))) float A1[T * S];
  __shared__ __attribute__((aligned(16))) float A2[T * S];
  __shared__ __attribute__((aligned(16))) float B1[T * S];
  __shared__ __attribute__((aligned(16))) float B2[T * S];
  __shared__ __attribute__((aligned(16))) float C1[T * CP];
  __shared__ __attribute__((aligned(16))) float C2[T * CP];
  __shared__ __attribute__((aligned(16))) float C3[T * CP];
  const int nt = n_tiles(n);
  const int tt = blockIdx.x % nt;
  const int split = blockIdx.x / nt;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int64_t t0 = (int64_t)tt * T;
  stage_tile<D>(A1, n1, t0, n);
  stage_tile<D>(A2, n2, t0, n);
  const int64_t gt0 = t0 + ty, gt1 = t0 + ty + 16;
  const float ct0 = gt0 < n ? c[gt0] : 0.f;
  const float ct1 = gt1 < n ? c[gt1] : 0.f;
  float g1[2][FPT], g2[2][FPT];
#pragma unroll
  for (int f = 0; f < FPT; ++f) g1[0][f] = g1[1][f] = g2[0][f] = g2[1][f] = 0.f;
  const int per = (nt + cs - 1) / cs;
  const int s_beg = split * per, s_end = min(nt, s_beg + per);
  for (int st = s_beg; st < s_end; ++st) {
    const int64_t s0 = (int64_t)st * T;
    __syncthreads();                 // previous tile's B*/C* fully consumed (and A* staged)
    stage_tile<D>(B1, n1, s0, n);
    stage_tile<D>(B2, n2, s0, n);
    __syncthreads();
    float s11[4], s12[4], s21[4];
    dot_2x2<D>(A1, B1, ty, tx, s11);   // n1_t . n1_s
    dot_2x2<D>(A1, B2, ty, tx, s12);   // n1_t . n2_s
    dot_2x2<D>(A2, B1, ty, tx, s21);   // n2_t . n1_s
    const int64_t gs0 = s0 + tx, gs1 = s0 + tx + 16;
    const float cs0 = gs0 < n ? c[gs0] : 0.f;
    const float cs1 = gs1 < n ? c[gs1] : 0.f;
    // micro-tile order: [0]=(t0,s0) [1]=(t0,s1) [2]=(t1,s0) [3]=(t1,s1)
    const bool v00 = gt0 < n && gs0 < n, v01 = gt0 < n && gs1 < n;
    const bool v10 = gt1 < n && gs0 < n, v11 = gt1 < n && gs1 < n;
    C1[ty * CP + tx] = v00 ? -ct0 * expf(s12[0] / tau) : 0.f;
    C1[ty * CP + tx + 16] = v01 ? -ct0 * expf(s12[1] / tau) : 0.f;
    C1[(ty + 16) * CP + tx] = v10 ? -ct1 * expf(s12[2] / tau) : 0.f;
    C1[(ty + 16) * CP + tx + 16] = v11 ? -ct1 * expf(s12[3] / tau) : 0.f;
    C2[ty * CP + tx] = (v00 && gt0 != gs0) ? -(ct0 + cs0) * expf(s11[0] / tau) : 0.f;
    C2[ty * CP + tx + 16] = (v01 && gt0 != gs1) ? -(ct0 + cs1) * expf(s11[1] / tau) : 0.f;
    C2[(ty + 16) * CP + tx] = (v10 && gt1 != gs0) ? -(ct1 + cs0) * expf(s11[2] / tau) : 0.f;
    C2[(ty + 16) * CP + tx + 16] = (v11 && gt1 != gs1) ? -(ct1 + cs1) * expf(s11[3] / tau) : 0.f;
    C3[ty * CP + tx] = v00 ? -cs0 * expf(s21[0] / tau) : 0.f;
    C3[ty * CP + tx + 16] = v01 ? -cs1 * expf(s21[1] / tau) : 0.f;
    C3[(ty + 16) * CP + tx] = v10 ? -cs0 * expf(s21[2] / tau) : 0.f;
    C3[(ty + 16) * CP + tx + 16] = v11 ? -cs1 * expf(s21[3] / tau) : 0.f;
    __syncthreads();
    // accumulate: rows {ty, ty+16}, features [tx*FPT, tx*FPT+FPT)
    //   g1[r] += C1[r][s]*n2_s + C2[r][s]*n1_s ;  g2[r] += C3[r][s]*n1_s
#pragma unroll 2
    for (int s = 0; s < T; s += 4) {
      const float4 p0 = *reinterpret_cast<const float4*>(C1 + ty * CP + s);
      const float4 p1 = *reinterpret_cast<const float4*>(C1 + (ty + 16) * CP + s);
      const float4 q0 = *reinterpret_cast<const float4*>(C2 + ty * CP + s);
      const float4 q1 = *reinterpret_cast<const float4*>(C2 + (ty + 16) * CP + s);
      const float4 r0 = *reinterpret_cast<const float4*>(C3 + ty * CP + s);
      const float4 r1 = *reinterpret_cast<const float4*>(C3 + (ty + 16) * CP + s);
      const float pc0[4] = {p0.x, p0.y, p0.z, p0.w}, pc1[4] = {p1.x, p1.y, p1.z, p1.w};
      const float qc0[4] = {q0.x, q0.y, q0.z, q0.w}, qc1[4] = {q1.x, q1.y, q1.z, q1.w};
      const float rc0[4] = {r0.x, r0.y, r0.z, r0.w}, rc1[4] = {r1.x, r1.y, r1.z, r1.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* b1 = B1 + (s + u) * S + tx * FPT;
        const float* b2 = B2 + (s + u) * S + tx * FPT;
#pragma unroll
        for (int f = 0; f < FPT; ++f) {
          const float x1 = b1[f], x2 = b2[f];
          g1[0][f] = fmaf(pc0[u], x2, fmaf(qc0[u], x1, g1[0][f]));
          g1[1][f] = fmaf(pc1[u], x2, fmaf(qc1[u], x1, g1[1][f]));
          g2[0][f] = fmaf(rc0[u], x1, g2[0][f]);
          g2[1][f] = fmaf(rc1[u], x1, g2[1][f]);
        }
      }
    }
  }
  const size_t base = (size_t)split * n * D;
  if (gt0 < n) {
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      g1p[base + gt0 * D + tx * FPT + f] = g1[0][f];
      g2p[base + gt0 * D + tx * FPT + f] = g2[0][f];
    }
  }
  if (gt1 < n) {
#pragma unroll
    for (int f = 0; f < FPT; ++f) {
      g1p[base + gt1 * D + tx * FPT + f] = g1[1][f];
      g2p[base + gt1 * D + tx * FPT + f] = g2[1][f];
    }
  }
}

// ---- backward finish: sum splits, diagonal terms, normalise-backward, scale by gloss ----------
//   z -> nrm = z*inv: gz = inv*(gn - nrm*(nrm.gn))  if |z| >= eps, else gn/eps
// With a gather index the rows are scatter-ADDED (hardware fp32 atomics) into the caller-zeroed
// table gradients; a NULL gz pointer skips that side (input that needs no gradient).
__device__ __forceinline__ void put_row(float* __restrict__ g, int64_t row, int d, int k, float4 v, bool scatter) {
  float* p = g + row * d + k * 4;
  if (scatter) {
    unsafeAtomicAdd(p + 0, v.x);
    unsafeAtomicAdd(p + 1, v.y);
    unsafeAtomicAdd(p + 2, v.z);
    unsafeAtomicAdd(p + 3, v.w);
  } else {
    *reinterpret_cast<float4*>(p) = v;
  }
}

__global__ __launch_bounds__(kBlock) void bwd_finish_kernel(const float* __restrict__ ws, size_t ws_stride, Layout L,
                                                            int P, int cs, const int64_t* __restrict__ idx,
                                                            int64_t n, int d, float tau,
                                                            const float* __restrict__ gloss, Z1Ptrs Z,
                                                            float* __restrict__ gz2, int finish_blocks,
                                                            float* __restrict__ losses, BprStepArgs bpr) {
  // Guest block (the one behind the finish blocks): the ASSEMBLY part of the hot step's BPR tail - the rows part ran
  // as guests of the prep launch. With deferred row terms (losses != NULL) the InfoNCE losses are reduced here first,
  // from the per-row-tile partials the backward tiles left (fixed order: a block-wide tree over the tiles); thread 0
  // writes them and thread 0 reads them back as terms of the assembly.
  if ((int)blockIdx.x >= finish_blocks) {
    if (losses) {
      __shared__ float red[4];
      for (int p = 0; p < P; ++p) {
        const float* __restrict__ part = ws + (size_t)p * ws_stride + L.loss;
        float acc = 0.f;
        for (int k = threadIdx.x; k < n_tiles(n); k += kBlock) acc += part[k];
        const float tt = block_sum_256(acc, red);
        if (threadIdx.x == 0) losses[p] = tt / (float)n;
      }
    }
    bpr_assemble_block(bpr);
    return;
  }
  constexpr int MAXC = 4;                      // d <= 256 -> at most 4 float4 chunks per lane
  const int lig = threadIdx.x & 15;
  const int64_t r = (int64_t)blockIdx.x * (kBlock / 16) + threadIdx.x / 16;
  if (r >= n) return;
  const int nch = d >> 2;
  const int64_t dst = idx ? idx[r] : r;
  const bool scatter = idx != nullptr;
  float4 vsum[MAXC];                           // gz2 of this row, summed over the problems
#pragma unroll
  for (int t = 0; t < MAXC; ++t) vsum[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int p = 0; p < P; ++p) {
    const float* __restrict__ wsp = ws + (size_t)p * ws_stride;
    const float* __restrict__ n1 = wsp + L.n1;
    const float* __restrict__ n2 = wsp + L.n2;
    const float* __restrict__ g1p = wsp + L.g1p;
    const float* __restrict__ g2p = wsp + L.g2p;
    const float gl = gloss[p];
    const float wt = wsp[L.w + r] / tau;
    float4 u[MAXC], v[MAXC], a[MAXC], b[MAXC];
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int t = 0; t < MAXC; ++t) {
      const int k = lig + 16 * t;
      if (k < nch) {
        a[t] = reinterpret_cast<const float4*>(n1 + r * d)[k];
        b[t] = reinterpret_cast<const float4*>(n2 + r * d)[k];
        float4 uu = make_float4(wt * b[t].x, wt * b[t].y, wt * b[t].z, wt * b[t].w);
        float4 vv = make_float4(wt * a[t].x, wt * a[t].y, wt * a[t].z, wt * a[t].w);
        // split partials: eight at a time, all sixteen loads in flight before the first add (one L2 round trip
        // per group instead of one per split; the order of the adds stays s = 0, 1, ...)
        int s = 0;
        for (; s + 8 <= cs; s += 8) {
          float4 x[8], y[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            x[e] = reinterpret_cast<const float4*>(g1p + ((size_t)(s + e) * n + r) * d)[k];
            y[e] = reinterpret_cast<const float4*>(g2p + ((size_t)(s + e) * n + r) * d)[k];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            uu.x += x[e].x; uu.y += x[e].y; uu.z += x[e].z; uu.w += x[e].w;
            vv.x += y[e].x; vv.y += y[e].y; vv.z += y[e].z; vv.w += y[e].w;
          }
        }
        for (; s < cs; ++s) {
          const float4 x = reinterpret_cast<const float4*>(g1p + ((size_t)s * n + r) * d)[k];
          const float4 y = reinterpret_cast<const float4*>(g2p + ((size_t)s * n + r) * d)[k];
          uu.x += x.x; uu.y += x.y; uu.z += x.z; uu.w += x.w;
          vv.x += y.x; vv.y += y.y; vv.z += y.z; vv.w += y.w;
        }
        u[t] = uu;
        v[t] = vv;
        p1 += f4_dot(a[t], uu);
        p2 += f4_dot(b[t], vv);
      }
    }
    p1 = group_sum<16>(p1);
    p2 = group_sum<16>(p2);
    const float i1 = wsp[L.inv1 + r], i2 = wsp[L.inv2 + r];
    // inv == 1/eps exactly when the row norm was clamped (|z| < eps): no projection term then
    const float q1 = (i1 >= 1.f / kNormEps) ? 0.f : p1;
    const float q2 = (i2 >= 1.f / kNormEps) ? 0.f : p2;
    const float m1 = gl * i1, m2 = gl * i2;
    float* __restrict__ gz1 = Z.gz1[p];
#pragma unroll
    for (int t = 0; t < MAXC; ++t) {
      const int k = lig + 16 * t;
      if (k < nch) {
        if (gz1)
          put_row(gz1, dst, d, k, make_float4(m1 * (u[t].x - a[t].x * q1), m1 * (u[t].y - a[t].y * q1),
                                              m1 * (u[t].z - a[t].z * q1), m1 * (u[t].w - a[t].w * q1)), scatter);
        vsum[t].x += m2 * (v[t].x - b[t].x * q2);
        vsum[t].y += m2 * (v[t].y - b[t].y * q2);
        vsum[t].z += m2 * (v[t].z - b[t].z * q2);
        vsum[t].w += m2 * (v[t].w - b[t].w * q2);
      }
    }
  }
  if (gz2) {
#pragma unroll
    for (int t = 0; t < MAXC; ++t) {
      const int k = lig + 16 * t;
      if (k < nch) put_row(gz2, dst, d, k, vsum[t], scatter);
    }
  }
}


// =================================================================================================
// MFMA tile kernels (D in {32, 64}): the n x n similarity tiles and, in the backward, the
// coefficient-tile x embedding products are fp32 matrix-core work (v_mfma_f32_32x32x2_f32).
//
// MFMA fragment layout of a [rows, D] row-major operand: lane l owns row (l & 31) and the D/2 features
// [h*D/2, (h+1)*D/2), h = l >> 5; MFMA step s then contracts features {s, D/2 + s} (any pairing of k indices is valid
// as long as A and B agree). Loading that shape straight from global (16 B per lane, 64 cache lines per instruction:
// the first form of these kernels) kept the CU's L1 busy for longer than the tile's MFMAs take - backward tiles 29.3 us,
// forward rows 19.9 us against 20.1 / 17.0 us with the LDS tile images below (n = 1024, two problems).
//
// C/D layout of the 32x32 MFMA: lane l holds C[(r & 3) + 8 (r >> 2) + 4 h][l & 31], r = 0..15.
// =================================================================================================
typedef float floatx16 __attribute__((ext_vector_type(16)));

// e^x for the similarity logits (|x| <= 1/tau: cosines): v_exp_f32 on x*log2(e), ~1 ulp; the full-range
// expf costs as many VALU cycles per tile as the MFMAs that produced it.
__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }

// ---- tile images in LDS (D in {32, 64}) -----------------------------------------------------------------------------
// A 32 x D tile of a row-major [rows, D] matrix is 32 D contiguous floats: it goes to LDS in whole lines by LDS-DMA
// (1 KB per wave instruction, lane-linear destination) and the fragments come from there. Image layout: row r at r * D floats, its 16-byte slots XOR-swizzled with the row number so
// that ds_read_b128 with lane = row is conflict-free (its 16-lane groups hold rows distinct mod 16); the swizzle goes
// on the SOURCE address of the DMA.
template <int D>
struct TileImg {
  static constexpr int SL = D / 4;                  // 16-byte slots per row
  static constexpr int RB = 64 / D;                 // rows per 256-byte bank row (1 or 2)
  static constexpr int RPP = 256 / D;               // rows per 1 KB DMA piece
  static constexpr int PIECES = T / RPP;
  static constexpr int FLOATS = T * D;
  __device__ static __forceinline__ int key(int row) { return (row / RB) & (SL - 1); }
  // whole wave: tile rows [0, 32) at src -> image at LDS byte address dst
  __device__ static __forceinline__ void stage(const float* __restrict__ src, unsigned dst, int lane) {
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const int row = p * RPP + lane / SL;
      const int slot = (lane % SL) ^ key(row);
      glds16(src + row * D + slot * 4, dst + (unsigned)p * 1024u);
    }
  }
  // one of several waves sharing the job: pieces p = first, first + step, ...
  __device__ static __forceinline__ void stage_part(const float* __restrict__ src, unsigned dst, int lane, int first, int step) {
    for (int p = first; p < PIECES; p += step) {
      const int row = p * RPP + lane / SL;
      const int slot = (lane % SL) ^ key(row);
      glds16(src + row * D + slot * 4, dst + (unsigned)p * 1024u);
    }
  }
  // MFMA fragment of lane (h, lr): row lr, features [h D/2, (h + 1) D/2)
  __device__ static __forceinline__ void frag(const float* __restrict__ img, int lr, int h, float (&f)[D / 2]) {
    const int k = key(lr);
#pragma unroll
    for (int q = 0; q < D / 8; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(img + lr * D + (((h * (SL / 2) + q) ^ k) << 2));
      f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w;
    }
  }
  // element (row, col) - as a B operand: one row per half-wave, col = 32 f + lr: conflict-free ds_read_b32
  __device__ static __forceinline__ float at(const float* __restrict__ img, int row, int col) {
    return img[row * D + ((((col >> 2) ^ key(row))) << 2) + (col & 3)];
  }
};

// ---- forward with LDS tile images: partial denominators per (row, block split). Similarity tiles are computed
// transposed (A operand = column tile, B operand = the block's row tile): accumulator register r of lane (h, lr) is
// the entry [row lr][column j(r, h)], so a row's partial sum is 16 in-register adds + one exchange between the
// half-waves instead of a 32-lane shuffle tree per register. The 4 waves of a block share the row tile, take column
// tiles round-robin (two per trip, both chains interleaved) and add their row sums through LDS in a fixed order.
template <int D>
__global__ __launch_bounds__(kBlock) void fwd_tiles_lds_kernel(const float* __restrict__ ws, size_t ws_stride,
                                                               Layout L, int64_t n, float tau, int cs) {
  using Img = TileImg<D>;
  constexpr int TF = Img::FLOATS;
  extern __shared__ __attribute__((aligned(16))) float lds[];         // [n1 row tile] [wave: two column tiles] x 4
  __shared__ float red[4][T];
  const float* __restrict__ wsp = ws + (size_t)blockIdx.y * ws_stride;
  const float* __restrict__ n1 = wsp + L.n1;
  const float* __restrict__ n2 = wsp + L.n2;
  float* __restrict__ rows_part = const_cast<float*>(wsp) + L.rows_part;
  const int nt = n_tiles(n);
  const int ti = blockIdx.x % nt;
  const int split = blockIdx.x / nt;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, lr = lane & 31;
  const int64_t i0 = (int64_t)ti * T;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
  const float* __restrict__ C0 = lds + TF + wave * 2 * TF;
  const float* __restrict__ C1 = C0 + TF;
  const unsigned c_lds = lds0 + (unsigned)(TF + wave * 2 * TF) * 4u;
  const int per = (2 * nt + cs - 1) / cs;
  const int c_beg = split * per, c_end = min(2 * nt, c_beg + per);
  const float inv_tau = 1.f / tau;
  // column tile ct of the walk: the first nt are tiles of n1 (the reflexive block, diagonal excluded), then n2
  auto stage_trip = [&](int ct) {
    Img::stage((ct < nt ? n1 : n2) + (int64_t)(ct < nt ? ct : ct - nt) * T * D, c_lds, lane);
    if (ct + 4 < c_end) {
      const int c2 = ct + 4;
      Img::stage((c2 < nt ? n1 : n2) + (int64_t)(c2 < nt ? c2 : c2 - nt) * T * D, c_lds + (unsigned)TF * 4u, lane);
    }
  };
  Img::stage_part(n1 + i0 * D, lds0, lane, wave, 4);
  int ct = c_beg + wave;
  if (ct < c_end) stage_trip(ct);
  vm_wait_n<0>();
  __syncthreads();                       // the row tile's pieces come from all four waves
  float a[D / 2];
  Img::frag(lds, lr, h, a);
  float rowsum = 0.f;
  const int own = (int)(i0 + lr);        // this lane's row (n < 2^31 on this path: the tile grid is 32-bit)
  auto accumulate = [&](const floatx16& acc, int ctile) {
    const bool refl = ctile < nt;
    const int j0 = (refl ? ctile : ctile - nt) * T;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int col = j0 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float e = fast_exp(acc[r] * inv_tau);
      rowsum += (col < (int)n && !(refl && col == own)) ? e : 0.f;
    }
  };
  for (; ct < c_end; ct += 8) {
    const bool two = ct + 4 < c_end;     // wave-uniform
    floatx16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    float b0[D / 2];
    Img::frag(C0, lr, h, b0);
    if (two) {
      float b1[D / 2];
      Img::frag(C1, lr, h, b1);
#pragma unroll
      for (int k = 0; k < D / 2; ++k) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[k], a[k], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[k], a[k], acc1, 0, 0, 0);
      }
      accumulate(acc0, ct);
      accumulate(acc1, ct + 4);
    } else {
#pragma unroll
      for (int k = 0; k < D / 2; ++k) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[k], a[k], acc0, 0, 0, 0);
      accumulate(acc0, ct);
    }
    if (ct + 8 < c_end) {                // next trip's images into this wave's own region
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      stage_trip(ct + 8);
      vm_wait_n<0>();
    }
  }
  rowsum += __shfl_xor(rowsum, 32, kWave);
  if (lane < T) red[wave][lane] = rowsum;
  __syncthreads();
  if (threadIdx.x < T) {
    const int64_t gi = i0 + threadIdx.x;
    if (gi < n)
      rows_part[(size_t)split * n + gi] =
          ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
  }
}
template <int D>
constexpr int fwd_tiles_lds_bytes() { return 9 * TileImg<D>::FLOATS * 4; }

// ---- backward pair tiles. Per (t, s) tile pair and wave, with the similarity tiles computed TRANSPOSED
// (A operand = tile s, B operand = tile t) so that accumulator register r of lane (h, lr) holds the entry
// [i = lr][j = j(r, h)], j(r, h) = (r & 3) + 8 (r >> 2) + 4 h - which is exactly an A operand of the second stage if
// MFMA step r contracts the rows {j(r, 0), j(r, 1)} of tile s (any pairing of k indices is valid as long as A and B
// agree): the coefficient tiles never leave the registers.
//   S12 = n1_t.n2_s^T, S11 = n1_t.n1_s^T, S21 = n2_t.n1_s^T            (3 x D/2 MFMAs)
//   C1 = -c_t e^{S12/tau}, C2 = -(c_t+c_s) e^{S11/tau} [t != s], C3 = -c_s e^{S21/tau}
//   g1_t += C1 . n2_s + C2 . n1_s ;  g2_t += C3 . n1_s                 (3 x 16 x D/32 MFMAs)
// LDS: the images of n1_t | n2_t (shared by the block) and of n1_s | n2_s per wave; second-stage B operands are rows
// of the s images. The 4 waves of a block share t, take s tiles round-robin and add their g1/g2 tiles in a fixed
// order through LDS before one store per block.
template <int D>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(2))) void bwd_tiles_lds_kernel(
    float* __restrict__ ws, size_t ws_stride, Layout L, int64_t n, float tau, int cs) {
  using Img = TileImg<D>;
  constexpr int TF = Img::FLOATS;
  constexpr int FT = D / 32;
  // [n1_t | n2_t] [wave: n1_s | n2_s] x 4 = 80 KB at D = 64: exactly two blocks per CU, so NO static LDS in this kernel
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ wsp = ws + (size_t)blockIdx.y * ws_stride;
  const float* __restrict__ n1 = wsp + L.n1;
  const float* __restrict__ n2 = wsp + L.n2;
  const float* __restrict__ c = wsp + L.c;
  float* __restrict__ g1p = wsp + L.g1p;
  float* __restrict__ g2p = wsp + L.g2p;
  const int nt = n_tiles(n);
  const int tt = blockIdx.x % nt;
  const int split = blockIdx.x / nt;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, lr = lane & 31;
  const int64_t t0 = (int64_t)tt * T;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds);
  const float* __restrict__ T1 = lds;
  const float* __restrict__ T2 = lds + TF;
  const float* __restrict__ S1 = lds + 2 * TF + wave * 2 * TF;
  const float* __restrict__ S2 = S1 + TF;
  const unsigned s_lds = lds0 + (unsigned)(2 * TF + wave * 2 * TF) * 4u;
  const int per = (nt + cs - 1) / cs;
  const int s_beg = split * per, s_end = min(nt, s_beg + per);
  // the block's t images: each wave moves a quarter of the pieces; its own first s images right behind
  Img::stage_part(n1 + t0 * D, lds0, lane, wave, 4);
  Img::stage_part(n2 + t0 * D, lds0 + (unsigned)TF * 4u, lane, wave, 4);
  int st = s_beg + wave;
  if (st < s_end) {
    Img::stage(n2 + (int64_t)st * T * D, s_lds + (unsigned)TF * 4u, lane);
    Img::stage(n1 + (int64_t)st * T * D, s_lds, lane);
  }
  const bool vi = t0 + lr < n;
  // Row terms deferred by the forward chain (meta[1]): c comes from the partial denominators right here - the
  // arithmetic of finalize_rows_kernel, which then was not launched - and the split-0 block of a row tile also leaves
  // w / c (for the finish) and the tile's loss partial.
  const bool defer = wsp[L.meta + 1] != 0.f;
  const float log_eps = wsp[L.meta];
  const float* __restrict__ rows_part = wsp + L.rows_part;
  const float* __restrict__ pos = wsp + L.pos;
  float ct = 0.f;
  if (defer) {
    float wi = 0.f, li = 0.f;
    if (vi) ct = row_terms(rows_part, pos, L.cs_f, n, t0 + lr, tau, log_eps, wi, li);
    if (split == 0 && wave == 0) {
      if (vi && h == 0) {
        wsp[L.w + t0 + lr] = wi;
        wsp[L.c + t0 + lr] = ct;
      }
      li = group_sum<32>(li);                  // both half-waves hold the same 32 rows
      if (lane == 0) wsp[L.loss + tt] = li;
    }
  } else if (vi) {
    ct = c[t0 + lr];
  }
  // deferred: c of row (s tile's first row + lr) for the coming trip, requested together with that trip's images
  auto s_terms = [&](int st_) {
    const int64_t i = (int64_t)st_ * T + lr;
    float wi, li;
    return i < n ? row_terms(rows_part, pos, L.cs_f, n, i, tau, log_eps, wi, li) : 0.f;
  };
  float cl_next = (defer && st < s_end) ? s_terms(st) : 0.f;
  floatx16 g1[FT], g2[FT];
#pragma unroll
  for (int f = 0; f < FT; ++f)
#pragma unroll
    for (int r = 0; r < 16; ++r) g1[f][r] = g2[f][r] = 0.f;
  const float inv_tau = 1.f / tau;
  vm_wait_n<0>();
  __syncthreads();                       // t images complete (pieces of all four waves)
  for (; st < s_end; st += 4) {
    const int64_t s0 = (int64_t)st * T;
    // column-side coefficients; everything past n is masked with 32-bit selects (c is padded to whole tiles, its pad
    // entries are never written; the pad rows of n1 / n2 are zeros)
    const int nj = (int)(n - s0 < T ? n - s0 : T);
    const int dj = tt == st ? lr : -1;             // the column that is this lane's own row (S11's diagonal)
    float cs_[16];
    if (defer) {                                   // wave-uniform
#pragma unroll
      for (int r = 0; r < 16; ++r) cs_[r] = __shfl(cl_next, (r & 3) + 8 * (r >> 2) + 4 * h, kWave);     // 0 past n
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float v = c[s0 + j];
        cs_[r] = j < nj ? v : 0.f;
      }
    }
    floatx16 s12, s11, s21;
#pragma unroll
    for (int r = 0; r < 16; ++r) s12[r] = s11[r] = s21[r] = 0.f;
    {
      float a1[D / 2], b2[D / 2], b1[D / 2];
      Img::frag(T1, lr, h, a1);
      Img::frag(S2, lr, h, b2);
      Img::frag(S1, lr, h, b1);
#pragma unroll
      for (int k = 0; k < D / 2; ++k) {          // two independent accumulation chains, interleaved
        s12 = __builtin_amdgcn_mfma_f32_32x32x2f32(b2[k], a1[k], s12, 0, 0, 0);
        s11 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[k], a1[k], s11, 0, 0, 0);
      }
      float a2[D / 2];
      Img::frag(T2, lr, h, a2);
#pragma unroll
      for (int k = 0; k < D / 2; ++k) s21 = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[k], a2[k], s21, 0, 0, 0);
    }
    // second-stage B operands (rows j(r, h) of the s images), requested while the last similarity MFMAs run: the
    // fragment registers are dead by now
    float q2[16][FT], q1[16][FT];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        q2[r][f] = Img::at(S2, j, 32 * f + lr);
        q1[r][f] = Img::at(S1, j, 32 * f + lr);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = (r & 3) + 8 * (r >> 2) + 4 * h;
      const float e12 = fast_exp(s12[r] * inv_tau), e11 = fast_exp(s11[r] * inv_tau), e21 = fast_exp(s21[r] * inv_tau);
      const float ctj = j < nj ? ct : 0.f;          // ct is 0 for rows past n
      const float c2 = (j < nj && j != dj) ? ct + cs_[r] : 0.f;
      s12[r] = -ctj * e12;
      s11[r] = vi ? -c2 * e11 : 0.f;
      s21[r] = vi ? -cs_[r] * e21 : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int f = 0; f < FT; ++f) {
        g1[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(s12[r], q2[r][f], g1[f], 0, 0, 0);
        g2[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(s21[r], q1[r][f], g2[f], 0, 0, 0);
        g1[f] = __builtin_amdgcn_mfma_f32_32x32x2f32(s11[r], q1[r][f], g1[f], 0, 0, 0);
      }
    if (st + 4 < s_end) {                 // next s images of this wave (its own region: no block barrier needed)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      Img::stage(n2 + (int64_t)(st + 4) * T * D, s_lds + (unsigned)TF * 4u, lane);
      Img::stage(n1 + (int64_t)(st + 4) * T * D, s_lds, lane);
      if (defer) cl_next = s_terms(st + 4);
      vm_wait_n<0>();
    }
  }
  // block reduction: waves 1..3 park their tiles in LDS (over the images: all waves are done with them), wave 0 adds
  // them in the order 1, 2, 3
  __syncthreads();
  constexpr int GW = 2 * T * D;           // floats per wave: g1 then g2, [row][feature]
  if (wave > 0) {
    float* dst = lds + (wave - 1) * GW;
#pragma unroll
    for (int f = 0; f < FT; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        dst[row * D + 32 * f + lr] = g1[f][r];
        dst[T * D + row * D + 32 * f + lr] = g2[f][r];
      }
  }
  __syncthreads();
  if (wave == 0) {
    const size_t base = (size_t)split * n * D;
#pragma unroll
    for (int f = 0; f < FT; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int o = row * D + 32 * f + lr;
        const float v1 = ((g1[f][r] + lds[o]) + lds[GW + o]) + lds[2 * GW + o];
        const float v2 = ((g2[f][r] + lds[T * D + o]) + lds[GW + T * D + o]) + lds[2 * GW + T * D + o];
        const int64_t gt = t0 + row;
        if (gt < n) {
          g1p[base + gt * D + 32 * f + lr] = v1;
          g2p[base + gt * D + 32 * f + lr] = v2;
        }
      }
  }
}
template <int D>
constexpr int bwd_tiles_lds_bytes() { return 10 * TileImg<D>::FLOATS * 4; }

inline bool use_mfma(int d) { return d == 32 || d == 64; }     // wider rows: the fp32-VALU tile kernels

inline bool infonce_d_ok(int d) { return d == 32 || d == 64 || d == 128 || d == 256; }

}  // namespace

namespace {

int infonce_fwd_impl(const float* const* z1s, const float* z2, const int64_t* idx, int P, int64_t n, int d,
                     float tau, float* losses, void* workspace, size_t workspace_bytes, void* stream,
                     float log_eps = 1e-8f, int phases = 3, int* tickets = nullptr, const BprStepArgs* rows_guest = nullptr,
                     bool defer_rows = false) {
  if (P < 1 || P > kMaxProblems || n <= 0 || !z1s || !z2 || !losses || !(tau > 0.f)) return MMSSL_E_BADARG;
  if (!infonce_d_ok(d)) return MMSSL_E_UNSUPP;
  const Layout L = make_layout(n, d);
  if (!workspace || workspace_bytes < (size_t)P * L.total * sizeof(float)) return MMSSL_E_WORKSPACE;
  Z1Ptrs Z;
  for (int p = 0; p < kMaxProblems; ++p) {
    Z.z1[p] = p < P ? z1s[p] : nullptr;
    Z.gz1[p] = nullptr;
    if (p < P && (!z1s[p] || ((uintptr_t)z1s[p] & 15))) return MMSSL_E_BADARG;
  }
  if (((uintptr_t)z2 | (uintptr_t)workspace) & 15) return MMSSL_E_BADARG;
  float* ws = reinterpret_cast<float*>(workspace);
  hipStream_t s = as_stream(stream);
  const int nt = n_tiles(n);
  const int rb = 2 * nt;                    // 16 rows per block, whole tiles (pad rows are written as zeros)
  const int fb = (int)((n + kBlock - 1) / kBlock);
  if (!(phases & 1)) {                      // phase 2 alone: only the loss scalars from the row terms
    hipLaunchKernelGGL(finalize_loss_kernel, dim3(1, P), dim3(kBlock), 0, s, ws, L.total, L, fb, n, losses);
    MMSSL_LAUNCH_CHECK();
    return 0;
  }
  BprStepArgs G;
  if (rows_guest) {
    if (d != 32 && d != 64) return MMSSL_E_UNSUPP;
    G = *rows_guest;
  } else {
    G.n_blocks = 0;
  }
  if (defer_rows && !use_mfma(d)) return MMSSL_E_UNSUPP;
  hipLaunchKernelGGL(prep_kernel, dim3(rb + G.n_blocks, P), dim3(kBlock), 0, s, Z, z2, idx, n, d, ws, L.total, L, rb, log_eps,
                     defer_rows ? 1 : 0, G);
  MMSSL_LAUNCH_CHECK();
  const dim3 grid(nt * L.cs_f, P);
  if (use_mfma(d)) {
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&fwd_tiles_lds_kernel<64>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, fwd_tiles_lds_bytes<64>());
    if (attr != hipSuccess) return (int)attr;
    if (d == 32) hipLaunchKernelGGL((fwd_tiles_lds_kernel<32>), grid, dim3(kBlock), fwd_tiles_lds_bytes<32>(), s, ws, L.total, L, n, tau, L.cs_f);
    else hipLaunchKernelGGL((fwd_tiles_lds_kernel<64>), grid, dim3(kBlock), fwd_tiles_lds_bytes<64>(), s, ws, L.total, L, n, tau, L.cs_f);
  } else switch (d) {
    case 32: hipLaunchKernelGGL((fwd_tiles_kernel<32>), grid, dim3(kBlock), 0, s, ws, L.total, L, n, tau, L.cs_f); break;
    case 64: hipLaunchKernelGGL((fwd_tiles_kernel<64>), grid, dim3(kBlock), 0, s, ws, L.total, L, n, tau, L.cs_f); break;
    case 128: hipLaunchKernelGGL((fwd_tiles_kernel<128>), grid, dim3(kBlock), 0, s, ws, L.total, L, n, tau, L.cs_f); break;
    case 256: hipLaunchKernelGGL((fwd_tiles_kernel<256>), grid, dim3(kBlock), 0, s, ws, L.total, L, n, tau, L.cs_f); break;
  }
  MMSSL_LAUNCH_CHECK();
  if (defer_rows) return 0;                 // row terms and losses: left to the backward tiles / finish
  hipLaunchKernelGGL(finalize_rows_kernel, dim3(fb, P), dim3(kBlock), 0, s, ws, L.total, L, L.cs_f, n, tau, log_eps,
                     tickets, losses);
  MMSSL_LAUNCH_CHECK();
  if (tickets || !(phases & 2)) return 0;
  hipLaunchKernelGGL(finalize_loss_kernel, dim3(1, P), dim3(kBlock), 0, s, ws, L.total, L, fb, n, losses);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

int infonce_bwd_impl(const int64_t* idx, int P, int64_t n, int d, float tau, const float* gloss,
                     float* const* gz1s, float* gz2, void* workspace, size_t workspace_bytes, void* stream,
                     int phases = 3, const BprStepArgs* assemble_guest = nullptr, float* deferred_losses = nullptr) {
  if (P < 1 || P > kMaxProblems || n <= 0 || !gloss || !(tau > 0.f) || phases < 1 || phases > 3) return MMSSL_E_BADARG;
  if (!infonce_d_ok(d)) return MMSSL_E_UNSUPP;
  const Layout L = make_layout(n, d);
  if (!workspace || workspace_bytes < (size_t)P * L.total * sizeof(float)) return MMSSL_E_WORKSPACE;
  Z1Ptrs Z;
  bool any = gz2 != nullptr;
  for (int p = 0; p < kMaxProblems; ++p) {
    Z.z1[p] = nullptr;
    Z.gz1[p] = (p < P && gz1s) ? gz1s[p] : nullptr;
    any = any || Z.gz1[p];
    if ((uintptr_t)Z.gz1[p] & 15) return MMSSL_E_BADARG;
  }
  if (!any) return MMSSL_E_BADARG;
  if (((uintptr_t)gz2 | (uintptr_t)workspace) & 15) return MMSSL_E_BADARG;
  float* ws = reinterpret_cast<float*>(workspace);
  hipStream_t s = as_stream(stream);
  const int nt = n_tiles(n);
  const dim3 grid(nt * L.cs_b, P);
  if ((phases & 1) && use_mfma(d)) {
    const dim3 g2 = grid;
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&bwd_tiles_lds_kernel<64>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, bwd_tiles_lds_bytes<64>());
    if (attr != hipSuccess) return (int)attr;
    if (d == 32) hipLaunchKernelGGL((bwd_tiles_lds_kernel<32>), g2, dim3(kBlock), bwd_tiles_lds_bytes<32>(), s, ws, L.total, L, n, tau, L.cs_b);
    else hipLaunchKernelGGL((bwd_tiles_lds_kernel<64>), g2, dim3(kBlock), bwd_tiles_lds_bytes<64>(), s, ws, L.total, L, n, tau, L.cs_b);
    MMSSL_LAUNCH_CHECK();
  } else if (phases & 1) {
    switch (d) {
      case 32: hipLaunchKernelGGL((bwd_tiles_kernel<32>), grid, dim3(kBlock), 0, s, ws, L.total, L, n, tau, L.cs_b); break;
      case 64: hipLaunchKernelGGL((bwd_tiles_kernel<64>), grid, dim3(kBlock), 0, s, ws, L.total, L, n, tau, L.cs_b); break;
      case 128: hipLaunchKernelGGL((bwd_tiles_kernel<128>), grid, dim3(kBlock), 0, s, ws, L.total, L, n, tau, L.cs_b); break;
      case 256: hipLaunchKernelGGL((bwd_tiles_kernel<256>), grid, dim3(kBlock), 0, s, ws, L.total, L, n, tau, L.cs_b); break;
    }
    MMSSL_LAUNCH_CHECK();
  }
  if (!(phases & 2)) return 0;
  const int rb = (int)((n + 15) / 16);
  BprStepArgs AG;
  if (assemble_guest) AG = *assemble_guest;
  else AG.n_blocks = 0;
  hipLaunchKernelGGL(bwd_finish_kernel, dim3(rb + (assemble_guest ? 1 : 0)), dim3(kBlock), 0, s, ws, L.total, L, P, L.cs_b,
                     idx, n, d, tau, gloss, Z, gz2, rb, deferred_losses, AG);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" size_t mmssl_infonce_workspace_bytes(int64_t n, int d) {
  if (n <= 0 || !infonce_d_ok(d)) return 0;
  return make_layout(n, d).total * sizeof(float);
}

extern "C" size_t mmssl_infonce_multi_workspace_bytes(int n_problems, int64_t n, int d) {
  if (n_problems < 1 || n_problems > kMaxProblems) return 0;
  return (size_t)n_problems * mmssl_infonce_workspace_bytes(n, d);
}

extern "C" int mmssl_infonce_fwd_f32(const float* z1, const float* z2, const int64_t* idx, int64_t n, int d,
                                     float tau, float* loss, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  const float* z1s[1] = {z1};
  return infonce_fwd_impl(z1s, z2, idx, 1, n, d, tau, loss, workspace, workspace_bytes, stream);
}

// The same loss with a caller-chosen constant inside the logarithm: -log(b_ii / D_i + log_eps). The trainer's
// variant (main.py:244) uses 1e-8; Models.batched_contrastive_loss of MMSSL (Models.py:79-98) and of the MICRO
// baseline (MICRO/codes/Models.py:74-95) use none (log_eps = 0). The backward is mmssl_infonce_bwd_f32 (the
// forward leaves the per-row coefficients, which already contain the constant, in the workspace).
extern "C" int mmssl_infonce_fwd_eps_f32(const float* z1, const float* z2, const int64_t* idx, int64_t n, int d,
                                         float tau, float log_eps, float* loss, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (!(log_eps >= 0.f)) return MMSSL_E_BADARG;
  const float* z1s[1] = {z1};
  return infonce_fwd_impl(z1s, z2, idx, 1, n, d, tau, loss, workspace, workspace_bytes, stream, log_eps);
}

extern "C" int mmssl_infonce_bwd_f32(const int64_t* idx, int64_t n, int d, float tau, const float* gloss,
                                     float* gz1, float* gz2, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  float* gz1s[1] = {gz1};
  return infonce_bwd_impl(idx, 1, n, d, tau, gloss, gz1s, gz2, workspace, workspace_bytes, stream);
}

extern "C" int mmssl_infonce_multi_fwd_f32(const float* const* z1s, const float* z2, const int64_t* idx,
                                           int n_problems, int64_t n, int d, float tau, float* losses,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  return infonce_fwd_impl(z1s, z2, idx, n_problems, n, d, tau, losses, workspace, workspace_bytes, stream);
}

extern "C" int mmssl_infonce_multi_fwd_phase_f32(const float* const* z1s, const float* z2, const int64_t* idx,
                                                 int n_problems, int64_t n, int d, float tau, float* losses,
                                                 void* workspace, size_t workspace_bytes, int phases, void* stream) {
  if (phases < 1 || phases > 3) return MMSSL_E_BADARG;
  return infonce_fwd_impl(z1s, z2, idx, n_problems, n, d, tau, losses, workspace, workspace_bytes, stream, 1e-8f,
                          phases);
}

extern "C" int mmssl_infonce_multi_fwd_ticket_f32(const float* const* z1s, const float* z2, const int64_t* idx,
                                                  int n_problems, int64_t n, int d, float tau, float* losses,
                                                  void* workspace, size_t workspace_bytes, int* tickets,
                                                  void* stream) {
  if (!tickets) return MMSSL_E_BADARG;
  return infonce_fwd_impl(z1s, z2, idx, n_problems, n, d, tau, losses, workspace, workspace_bytes, stream, 1e-8f, 3,
                          tickets);
}

extern "C" int mmssl_infonce_multi_bwd_f32(const int64_t* idx, int n_problems, int64_t n, int d, float tau,
                                           const float* gloss, float* const* gz1s, float* gz2, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  return infonce_bwd_impl(idx, n_problems, n, d, tau, gloss, gz1s, gz2, workspace, workspace_bytes, stream);
}

extern "C" size_t mmssl_bpr_workspace_bytes(int64_t B);

// ---- the hot step's loss chain with the BPR tail in two parts (d in {32, 64}) ---------------------------------------
// As guest blocks of the backward PAIR TILES (an earlier form) the tail's B/16 blocks pushed that launch past one round
// of resident blocks (512 pair-tile blocks fill the chip at two blocks per CU: the last 64 wait for a slot), and a
// guest's static LDS would cost the pair tiles their second block per CU. Here the ROWS part rides in the short prep
// launch and the one-block ASSEMBLY part in the backward finish: same arithmetic, same bits; measured pair tiles
// 34.7 -> 29.4 us, prep 5.7 -> 8.4 us (profiles/r03/step_timeline_bpr_two_parts.txt).
extern "C" int mmssl_infonce_multi_fwd_ticket_bpr_f32(const float* const* z1s, const float* z2, const int64_t* idx,
                                                      int n_problems, int64_t n, int d, float tau, float* losses,
                                                      void* workspace, size_t workspace_bytes, int* tickets,
                                                      const float* Eu, const float* Ei, const int64_t* users,
                                                      const int64_t* pos, const int64_t* neg, int64_t B, float decay,
                                                      int64_t batch_size, const float* g_mf, const float* g_emb, float* gEu,
                                                      float* gEi, void* bpr_workspace, size_t bpr_workspace_bytes,
                                                      void* stream) {
  if (!tickets) return MMSSL_E_BADARG;
  if (!infonce_d_ok(d) || !use_mfma(d)) return MMSSL_E_UNSUPP;
  BprStepArgs A;
  float dummy_terms[4];
  int dummy_ticket = 0;
  // only the rows part's fields are used by this launch; the rest is validated by the finish entry
  const int rc = make_bpr_step_args(A, Eu, Ei, users, pos, neg, B, d, decay, batch_size, g_mf, g_emb, gEu, gEi, dummy_terms,
                                    dummy_terms, 3, nullptr, 0.f, dummy_terms, nullptr, 0, nullptr, 0, bpr_workspace,
                                    bpr_workspace_bytes, mmssl_bpr_workspace_bytes(B), &dummy_ticket, nullptr, 0);
  if (rc != 0) return rc;
  return infonce_fwd_impl(z1s, z2, idx, n_problems, n, d, tau, losses, workspace, workspace_bytes, stream, 1e-8f, 3,
                          tickets, &A, true);
}

extern "C" int mmssl_infonce_multi_bwd_finish_bpr_f32(const int64_t* idx, int n_problems, int64_t n, int d, float tau,
                                                      const float* gloss, float* losses, float* const* gz1s, float* gz2, void* workspace,
                                                      size_t workspace_bytes, const float* Eu, const float* Ei,
                                                      const int64_t* users, const int64_t* pos, const int64_t* neg,
                                                      int64_t B, float decay, int64_t batch_size, const float* g_mf,
                                                      const float* g_emb, float* gEu, float* gEi, float* terms,
                                                      const float* w, int n_terms, const float* extra, float c,
                                                      float* total, float* const* f32_ticks, int n_f32,
                                                      uint64_t* const* u64_ticks, int n_u64, void* bpr_workspace,
                                                      size_t bpr_workspace_bytes, const float* extra_parts,
                                                      int64_t n_extra_parts, void* stream) {
  if (!infonce_d_ok(d) || !use_mfma(d)) return MMSSL_E_UNSUPP;
  if (!losses) return MMSSL_E_BADARG;
  BprStepArgs A;
  int dummy_ticket = 0;
  const int rc = make_bpr_step_args(A, Eu, Ei, users, pos, neg, B, d, decay, batch_size, g_mf, g_emb, gEu, gEi, terms, w,
                                    n_terms, extra, c, total, f32_ticks, n_f32, u64_ticks, n_u64, bpr_workspace,
                                    bpr_workspace_bytes, mmssl_bpr_workspace_bytes(B), &dummy_ticket, extra_parts,
                                    n_extra_parts);
  if (rc != 0) return rc;
  return infonce_bwd_impl(idx, n_problems, n, d, tau, gloss, gz1s, gz2, workspace, workspace_bytes, stream, 2, &A, losses);
}

extern "C" int mmssl_infonce_multi_bwd_phase_f32(const int64_t* idx, int n_problems, int64_t n, int d, float tau,
                                                 const float* gloss, float* const* gz1s, float* gz2,
                                                 void* workspace, size_t workspace_bytes, int phases, void* stream) {
  return infonce_bwd_impl(idx, n_problems, n, d, tau, gloss, gz1s, gz2, workspace, workspace_bytes, stream, phases);
}
