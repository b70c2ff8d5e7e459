// Batch similarity rows and per-row top-K for gfx950 (MI355X).
//
//   sim rows : S[b, j] = < Q[qidx[b], :], T[j, :] >  for all items j, with the entries listed in a CSR row of the
//              batch user replaced by `mask_value` — the [B, d] x [d, n_items] score product of
//              Trainer.u_sim_calculation (/root/reference/MMSSL/main.py:283-298, mask value 0, followed by the row
//              normalisation) and of the evaluation (/root/reference/MMSSL/utility/batch_test.py:150-152 + the
//              training-item exclusion of :91-100, mask value -inf).
//              fp32 MFMA tiles (v_mfma_f32_32x32x2_f32: exact fp32): a wave owns 32 batch rows (A operand, kept in
//              registers for the whole launch) and walks 32-item tiles of its block's item chunk; the item rows
//              stream from L2 straight into MFMA fragment layout (16 B per lane, double buffered), the CSR mask
//              is a per-block bitmap in LDS (one 32-bit word per item = the 32 batch rows of the block), and the
//              row sums of squares for the normalisation leave the kernel as fixed-order partials: the [B, n_items]
//              matrix is written once and never re-read by a separate masking pass.
//   top-K    : the K best entries of every row in DESCENDING score, ties by ASCENDING item id — the order
//              heapq.nlargest gives the reference (batch_test.py:21-36). One block per row: the row sits in LDS as
//              order-preserving integer keys, the K-th key is found by a 32-step bit bisection (no sort of the
//              18 K columns), winners are compacted in id order and the <= 64 of them sorted by one wave.
#include "common.hpp"

using namespace mmssl;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kSimChunk = 2048;          // items per block (bitmap: 8 KB of LDS)
constexpr int kSimTile = 32;

__device__ __forceinline__ int32_t mask_col(const void* cols, int stride, int64_t e) {
  return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(cols) + e * stride);
}

// DCH = d / 8: number of 8-deep k chunks (one float4 per lane half)
template <int DCH>
__global__ __launch_bounds__(kBlock) void sim_tiles_kernel(const float* __restrict__ Q, const int64_t* __restrict__ qidx,
                                                           int64_t B, const float* __restrict__ T, int64_t I,
                                                           const int32_t* __restrict__ m_rowptr,
                                                           const void* __restrict__ m_cols, int m_stride,
                                                           float mask_value, float* __restrict__ out, int64_t ldo,
                                                           float* __restrict__ sumsq_part, int nparts) {
  constexpr int d = DCH * 8;
  __shared__ uint32_t bitmap[kSimChunk];
  __shared__ float red[4][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, kh = lane >> 5;
  const int64_t u0 = (int64_t)blockIdx.x * 32;
  const int64_t c0 = (int64_t)blockIdx.y * kSimChunk;
  const int64_t c1 = min(I, c0 + kSimChunk);
  for (int i = tid; i < kSimChunk; i += kBlock) bitmap[i] = 0u;
  __syncthreads();
  if (m_rowptr) {                          // thread (user ui, part): the user's masked items inside [c0, c1)
    const int ui = tid & 31, part = tid >> 5;
    if (u0 + ui < B) {
      const int64_t r = qidx ? qidx[u0 + ui] : (u0 + ui);
      int lo = m_rowptr[r], hi = m_rowptr[r + 1];
      const int end = hi;
      while (lo < hi) {                    // lower bound of c0 in the sorted column list
        const int mid = (lo + hi) >> 1;
        if (mask_col(m_cols, m_stride, mid) < c0) lo = mid + 1;
        else hi = mid;
      }
      for (int e = lo + part; e < end; e += 8) {
        const int32_t c = mask_col(m_cols, m_stride, e);
        if (c >= c1) break;
        atomicOr(&bitmap[c - c0], 1u << ui);
      }
    }
  }
  // A operand: my batch row (lane half kh holds k = 8q + 4kh .. +3 of every 8-chunk), zero past B
  float4 qf[DCH];
  {
    const int64_t ub = u0 + n;
    const float4* qrow = nullptr;
    if (ub < B) qrow = reinterpret_cast<const float4*>(Q + (qidx ? qidx[ub] : ub) * d);
#pragma unroll
    for (int q = 0; q < DCH; ++q) qf[q] = qrow ? qrow[2 * q + kh] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  float sq[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) sq[r] = 0.f;
  const int n_tiles = (int)((c1 - c0 + kSimTile - 1) / kSimTile);
  auto load_items = [&](int tile, float4 (&tf)[DCH]) {
    const int64_t j = min(c0 + (int64_t)tile * kSimTile + n, I - 1);
    const float4* trow = reinterpret_cast<const float4*>(T + j * d);
#pragma unroll
    for (int q = 0; q < DCH; ++q) tf[q] = trow[2 * q + kh];
  };
  float4 ta[DCH], tb[DCH];
  auto do_tile = [&](int tile, const float4 (&tf)[DCH]) {
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < DCH; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[q].x, tf[q].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[q].y, tf[q].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[q].z, tf[q].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[q].w, tf[q].w, acc, 0, 0, 0);
    }
    const int64_t j = c0 + (int64_t)tile * kSimTile + n;
    const uint32_t word = bitmap[tile * kSimTile + n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * kh;
      const bool masked = (word >> m) & 1u;
      const float v = masked ? mask_value : acc[r];
      if (u0 + m < B && j < I) {
        out[(u0 + m) * ldo + j] = v;
        if (!masked) sq[r] = fmaf(v, v, sq[r]);
      }
    }
  };
  int tile = wave;
  if (tile < n_tiles) load_items(tile, ta);
  for (; tile < n_tiles; tile += 8) {       // two tiles per iteration: register double buffer without copies
    if (tile + 4 < n_tiles) load_items(tile + 4, tb);
    do_tile(tile, ta);
    if (tile + 4 < n_tiles) {
      if (tile + 8 < n_tiles) load_items(tile + 8, ta);
      do_tile(tile + 4, tb);
    }
  }
  if (sumsq_part) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = sq[r];
#pragma unroll
      for (int mk = 1; mk < 32; mk <<= 1) v += __shfl_xor(v, mk, kWave);
      if (n == 0) red[wave][(r & 3) + 8 * (r >> 2) + 4 * kh] = v;
    }
    __syncthreads();
    if (tid < 32 && u0 + tid < B)
      sumsq_part[(u0 + tid) * nparts + blockIdx.y] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
  }
}

// X[b, :] *= 1 / max(sqrt(sum of the row's partials), eps)   (F.normalize(dim=1), main.py:297); inv_out[b] gets the factor
__global__ __launch_bounds__(kBlock) void rows_scale_parts_kernel(float* __restrict__ X, int64_t I, int64_t ldo,
                                                                  const float* __restrict__ part, int nparts, float eps,
                                                                  float* __restrict__ inv_out) {
  const int64_t b = blockIdx.x;
  float tot = 0.f;
  for (int k = 0; k < nparts; ++k) tot += part[b * nparts + k];          // fixed order, same in every thread
  const float inv = 1.f / fmaxf(sqrtf(tot), eps);
  float* __restrict__ row = X + b * ldo;
  for (int64_t j = threadIdx.x; j < I; j += kBlock) row[j] *= inv;
  if (threadIdx.x == 0 && inv_out) inv_out[b] = inv;
}

// ---------------------------------------------------------------------------------------------------------------
// per-row top-K
// ---------------------------------------------------------------------------------------------------------------
constexpr int kTopkMaxCols = 36864;        // 144 KB of keys in LDS
constexpr int kTopkMaxK = 256;             // K <= 64: one wave sorts the winners; up to 256: the block does

__device__ __forceinline__ uint32_t order_key(float x) {      // monotone: a < b  <=>  key(a) < key(b); -0 == +0 apart
  const uint32_t b = __float_as_uint(x);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__device__ __forceinline__ int block_sum_i32(int v, int* red) {          // every thread gets the total
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, kWave);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(kBlock) void topk_rows_kernel(const float* __restrict__ X, int64_t I, int64_t ldx, int K,
                                                           int64_t* __restrict__ idx_out, float* __restrict__ val_out) {
  extern __shared__ uint32_t keys[];                        // [I]
  __shared__ int red[4];
  __shared__ int scan_gt[kBlock], scan_eq[kBlock];
  __shared__ uint32_t win_key[kTopkMaxK];
  __shared__ int32_t win_idx[kTopkMaxK];
  static_assert(kTopkMaxK == kBlock, "the block-wide sort gives every thread one winner slot");
  const int tid = threadIdx.x;
  const int64_t b = blockIdx.x;
  const float* __restrict__ row = X + b * ldx;
  for (int64_t j = tid; j < I; j += kBlock) keys[j] = order_key(row[j]);
  __syncthreads();
  const int Ke = (int)min((int64_t)K, I);
  // K-th largest key: the largest t with #{keys >= t} >= Ke, one bit at a time
  uint32_t tau = 0u;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t cand = tau | (1u << bit);
    int c = 0;
    for (int64_t j = tid; j < I; j += kBlock) c += keys[j] >= cand ? 1 : 0;
    if (block_sum_i32(c, red) >= Ke) tau = cand;
  }
  // winners in ascending id order: everything above tau, then the first (Ke - #above) entries equal to tau
  const int L = (int)((I + kBlock - 1) / kBlock);
  const int64_t jb = (int64_t)tid * L, je = min(I, jb + L);
  int gt = 0, eq = 0;
  for (int64_t j = jb; j < je; ++j) {
    const uint32_t k = keys[j];
    gt += k > tau ? 1 : 0;
    eq += k == tau ? 1 : 0;
  }
  scan_gt[tid] = gt;
  scan_eq[tid] = eq;
  __syncthreads();
  if (tid == 0) {                           // 256-entry exclusive scans (serial: ~0.5 us, once per row)
    int a = 0, e = 0;
    for (int t = 0; t < kBlock; ++t) {
      const int x = scan_gt[t], y = scan_eq[t];
      scan_gt[t] = a;
      scan_eq[t] = e;
      a += x;
      e += y;
    }
    red[0] = a;
  }
  if (tid < kTopkMaxK) { win_key[tid] = 0u; win_idx[tid] = 0x7fffffff; }
  __syncthreads();
  const int n_gt = red[0];
  const int need_eq = Ke - n_gt;
  int pg = scan_gt[tid], pe = scan_eq[tid];
  for (int64_t j = jb; j < je; ++j) {
    const uint32_t k = keys[j];
    if (k > tau) {
      win_key[pg] = k;
      win_idx[pg] = (int32_t)j;
      ++pg;
    } else if (k == tau) {
      if (pe < need_eq) {
        win_key[n_gt + pe] = k;
        win_idx[n_gt + pe] = (int32_t)j;
      }
      ++pe;
    }
  }
  __syncthreads();
  if (K > 64) {                             // block-uniform: up to 256 winners, bitonic network through LDS
    uint32_t k = tid < Ke ? win_key[tid] : 0u;
    int32_t id = tid < Ke ? win_idx[tid] : 0x7fffffff;
    for (int size = 2; size <= kBlock; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        __syncthreads();
        win_key[tid] = k;
        win_idx[tid] = id;
        __syncthreads();
        const uint32_t ok = win_key[tid ^ stride];
        const int32_t oid = win_idx[tid ^ stride];
        const bool mine_first = (k > ok) || (k == ok && id < oid);
        const bool lower = (tid & stride) == 0;
        const bool asc = (tid & size) == 0;
        const bool keep_mine = (lower == asc) ? mine_first : !mine_first;
        if (!keep_mine) { k = ok; id = oid; }
      }
    }
    if (tid < Ke) {
      idx_out[b * K + tid] = id;
      if (val_out) val_out[b * K + tid] = key_value(k);
    } else if (tid < K) {
      idx_out[b * K + tid] = -1;
      if (val_out) val_out[b * K + tid] = 0.f;
    }
    return;
  }
  if (tid < 64) {                           // one wave sorts the <= 64 winners: key descending, id ascending
    uint32_t k = win_key[tid];
    int32_t id = win_idx[tid];
    const bool valid = tid < Ke;
    if (!valid) { k = 0u; id = 0x7fffffff; }
    // "a before b"  <=>  ka > kb or (ka == kb and ia < ib); bitonic network over 64 lanes
    for (int size = 2; size <= 64; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        const uint32_t ok = __shfl_xor(k, stride, kWave);
        const int32_t oid = __shfl_xor(id, stride, kWave);
        const bool mine_first = (k > ok) || (k == ok && id < oid);
        const bool lower = (tid & stride) == 0;                 // I keep the "first" of the pair in ascending blocks
        const bool asc = (tid & size) == 0;
        const bool keep_mine = (lower == asc) ? mine_first : !mine_first;
        if (!keep_mine) { k = ok; id = oid; }
      }
    }
    if (tid < Ke) {
      idx_out[b * K + tid] = id;
      if (val_out) val_out[b * K + tid] = key_value(k);
    } else if (tid < K) {
      idx_out[b * K + tid] = -1;
      if (val_out) val_out[b * K + tid] = 0.f;
    }
  }
}

// out[b, k] = 1 if cand[b, k] is in the (sorted) CSR row rows[b]
__global__ __launch_bounds__(kBlock) void rows_membership_kernel(const int32_t* __restrict__ rowptr,
                                                                 const int32_t* __restrict__ cols,
                                                                 const int64_t* __restrict__ rows, int64_t total, int K,
                                                                 const int64_t* __restrict__ cand,
                                                                 uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= total) return;
  const int64_t r = rows[i / K];
  const int64_t c = cand[i];
  int lo = rowptr[r], hi = rowptr[r + 1];
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cols[mid] < c) lo = mid + 1;
    else hi = mid;
  }
  out[i] = (lo < rowptr[r + 1] && cols[lo] == c) ? 1 : 0;
}


// d = 256 (embed_size 256): the same tiles with the reduction walked in two 128-deep halves; both operands of a half
// are read when the half starts (the batch rows stay in L1/L2), no register double buffer - a rare shape, kept exact.
__global__ __launch_bounds__(kBlock) void sim_tiles_wide_kernel(const float* __restrict__ Q, const int64_t* __restrict__ qidx,
                                                                int64_t B, const float* __restrict__ T, int64_t I,
                                                                const int32_t* __restrict__ m_rowptr,
                                                                const void* __restrict__ m_cols, int m_stride,
                                                                float mask_value, float* __restrict__ out, int64_t ldo,
                                                                float* __restrict__ sumsq_part, int nparts) {
  constexpr int d = 256, DCH = 16;           // chunks per half
  __shared__ uint32_t bitmap[kSimChunk];
  __shared__ float red[4][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, kh = lane >> 5;
  const int64_t u0 = (int64_t)blockIdx.x * 32;
  const int64_t c0 = (int64_t)blockIdx.y * kSimChunk;
  const int64_t c1 = min(I, c0 + kSimChunk);
  for (int i = tid; i < kSimChunk; i += kBlock) bitmap[i] = 0u;
  __syncthreads();
  if (m_rowptr) {
    const int ui = tid & 31, part = tid >> 5;
    if (u0 + ui < B) {
      const int64_t r = qidx ? qidx[u0 + ui] : (u0 + ui);
      int lo = m_rowptr[r], hi = m_rowptr[r + 1];
      const int end = hi;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (mask_col(m_cols, m_stride, mid) < c0) lo = mid + 1;
        else hi = mid;
      }
      for (int e = lo + part; e < end; e += 8) {
        const int32_t c = mask_col(m_cols, m_stride, e);
        if (c >= c1) break;
        atomicOr(&bitmap[c - c0], 1u << ui);
      }
    }
  }
  const int64_t ub = u0 + n;
  const float4* qrow = ub < B ? reinterpret_cast<const float4*>(Q + (qidx ? qidx[ub] : ub) * d) : nullptr;
  __syncthreads();
  float sq[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) sq[r] = 0.f;
  const int n_tiles = (int)((c1 - c0 + kSimTile - 1) / kSimTile);
  for (int tile = wave; tile < n_tiles; tile += 4) {
    const int64_t j = c0 + (int64_t)tile * kSimTile + n;
    const float4* trow = reinterpret_cast<const float4*>(T + min(j, I - 1) * d);
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 qf[DCH], tf[DCH];
#pragma unroll
      for (int q = 0; q < DCH; ++q) {
        qf[q] = qrow ? qrow[2 * (half * DCH + q) + kh] : make_float4(0.f, 0.f, 0.f, 0.f);
        tf[q] = trow[2 * (half * DCH + q) + kh];
      }
#pragma unroll
      for (int q = 0; q < DCH; ++q) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[q].x, tf[q].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[q].y, tf[q].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[q].z, tf[q].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[q].w, tf[q].w, acc, 0, 0, 0);
      }
    }
    const uint32_t word = bitmap[tile * kSimTile + n];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * kh;
      const bool masked = (word >> m) & 1u;
      const float v = masked ? mask_value : acc[r];
      if (u0 + m < B && j < I) {
        out[(u0 + m) * ldo + j] = v;
        if (!masked) sq[r] = fmaf(v, v, sq[r]);
      }
    }
  }
  if (sumsq_part) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = sq[r];
#pragma unroll
      for (int mk = 1; mk < 32; mk <<= 1) v += __shfl_xor(v, mk, kWave);
      if (n == 0) red[wave][(r & 3) + 8 * (r >> 2) + 4 * kh] = v;
    }
    __syncthreads();
    if (tid < 32 && u0 + tid < B)
      sumsq_part[(u0 + tid) * nparts + blockIdx.y] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
  }
}

}  // namespace

namespace mmssl {
// shared with graph.hip (mmssl_graph_sim_rows_f32: the mask is the plan's CSR, {col, val} pairs = stride 8)
int sim_launch(const float* Q, const int64_t* qidx, int64_t B, const float* T, int64_t I, int d, const int32_t* m_rowptr,
               const void* m_cols, int m_stride, float mask_value, float* out, int64_t ldo, float* sumsq_part,
               hipStream_t s) {
  const int nparts = (int)((I + kSimChunk - 1) / kSimChunk);
  const dim3 grid((unsigned)((B + 31) / 32), (unsigned)nparts);
#define SIM_CASE(DCH)                                                                                                   \
  hipLaunchKernelGGL((sim_tiles_kernel<DCH>), grid, dim3(kBlock), 0, s, Q, qidx, B, T, I, m_rowptr, m_cols, m_stride,   \
                     mask_value, out, ldo, sumsq_part, nparts)
  switch (d) {
    case 32: SIM_CASE(4); break;
    case 64: SIM_CASE(8); break;
    case 128: SIM_CASE(16); break;
    case 256:
      hipLaunchKernelGGL(sim_tiles_wide_kernel, grid, dim3(kBlock), 0, s, Q, qidx, B, T, I, m_rowptr, m_cols, m_stride,
                         mask_value, out, ldo, sumsq_part, nparts);
      break;
    default: return MMSSL_E_UNSUPP;
  }
#undef SIM_CASE
  MMSSL_LAUNCH_CHECK();
  return 0;
}
}  // namespace mmssl

extern "C" int mmssl_sim_rows_parts(int64_t n_items) {
  return n_items <= 0 ? 0 : (int)((n_items + kSimChunk - 1) / kSimChunk);
}

extern "C" int mmssl_sim_rows_f32(const float* Q, const int64_t* qidx, int64_t B, const float* T, int64_t n_items, int d,
                                  const int32_t* mask_rowptr, const int32_t* mask_cols, float mask_value, float* out,
                                  int64_t ldo, float* sumsq_part, void* stream) {
  if (B < 0 || n_items < 0 || ldo < n_items) return MMSSL_E_BADARG;
  if (B == 0 || n_items == 0) return 0;
  if (!Q || !T || !out || (mask_rowptr && !mask_cols)) return MMSSL_E_BADARG;
  if (((uintptr_t)Q | (uintptr_t)T) & 15) return MMSSL_E_BADARG;
  return sim_launch(Q, qidx, B, T, n_items, d, mask_rowptr, mask_cols, 4, mask_value, out, ldo, sumsq_part,
                    as_stream(stream));
}

extern "C" int mmssl_rows_scale_parts_f32(float* X, int64_t B, int64_t n_items, int64_t ldo, const float* sumsq_part,
                                          int nparts, float eps, float* inv_out, void* stream) {
  if (B < 0 || n_items < 0 || ldo < n_items || nparts < 1 || !(eps > 0.f)) return MMSSL_E_BADARG;
  if (B == 0 || n_items == 0) return 0;
  if (!X || !sumsq_part) return MMSSL_E_BADARG;
  hipLaunchKernelGGL(rows_scale_parts_kernel, dim3((unsigned)B), dim3(kBlock), 0, as_stream(stream), X, n_items, ldo,
                     sumsq_part, nparts, eps, inv_out);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_topk_rows_f32(const float* X, int64_t B, int64_t n_cols, int64_t ldx, int K, int64_t* idx_out,
                                   float* val_out, void* stream) {
  if (B < 0 || n_cols <= 0 || ldx < n_cols || K < 1 || !idx_out) return MMSSL_E_BADARG;
  if (K > kTopkMaxK || n_cols > kTopkMaxCols) return MMSSL_E_UNSUPP;
  if (B == 0) return 0;
  if (!X) return MMSSL_E_BADARG;
  static const int attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(topk_rows_kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, kTopkMaxCols * 4);
  if (attr != 0) return MMSSL_E_UNSUPP;
  hipLaunchKernelGGL(topk_rows_kernel, dim3((unsigned)B), dim3(kBlock), (size_t)n_cols * 4, as_stream(stream), X, n_cols,
                     ldx, K, idx_out, val_out);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_rows_membership_u8(const int32_t* rowptr, const int32_t* cols, const int64_t* rows, int64_t B, int K,
                                        const int64_t* cand, uint8_t* out, void* stream) {
  if (B < 0 || K < 1) return MMSSL_E_BADARG;
  if (B == 0) return 0;
  if (!rowptr || !rows || !cand || !out) return MMSSL_E_BADARG;
  const int64_t total = B * K;
  hipLaunchKernelGGL(rows_membership_kernel, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     as_stream(stream), rowptr, cols, rows, total, K, cand, out);
  MMSSL_LAUNCH_CHECK();
  return 0;
}
