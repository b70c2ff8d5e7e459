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
//              heapq.nlargest gives the reference (batch_test.py:21-36). One block per row: the row sits in REGISTERS as
//              order-preserving integer keys, the K-th key is found by a 32-step bit bisection (no sort of the
//              18 K columns), a tie at the cut is resolved towards the smaller ids by a second bisection over the
//              ids, and the <= 64 winners are sorted by one wave.
#include "common.hpp"

using namespace mmssl;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kSimChunk = 512;           // items per block: 36 x 32 blocks for the Baby shape (4.5 per CU: no tail round)
constexpr int kSimTile = 32;
constexpr int kSimGroup = 256;           // items staged in LDS and written out as whole 1 KB row segments
constexpr int kStagePitch = kSimGroup + 4;
static_assert(kSimChunk == 2 * kSimGroup, "a block stages its chunk as two groups, each in a buffer of its own");

__device__ __forceinline__ int32_t mask_col(const void* cols, int stride, int64_t e) {
  return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(cols) + e * stride);
}

// DCH = d / 8: number of 8-deep k chunks (one float4 per lane half)
// A block = 32 batch rows x 512 items: every wave computes four 32 x 32 MFMA tiles (item rows stream from L2 straight
// into fragment layout, register double buffer), leaves them - masked, scaled by the row's factor - in an LDS stage
// of 32 rows x 256 items, and the block then writes each staged row as ONE 1 KB segment (round 5 stored the MFMA
// fragments directly: 128-byte segments, 0.18 of the write roofline together with the separate scale pass).
// row_scale != NULL: out = row_scale[b] * score (the factor of F.normalize, known BEFORE this launch: usim_norms_kernel),
// so the [B, n_items] matrix is written exactly once. Columns [I, ldo) of a padded row are written as zeros.
template <int DCH>
__global__ __launch_bounds__(kBlock) void sim_tiles_kernel(const float* __restrict__ Q, const int64_t* __restrict__ qidx,
                                                           int64_t B, const float* __restrict__ T, int64_t I,
                                                           const int32_t* __restrict__ m_rowptr,
                                                           const void* __restrict__ m_cols, int m_stride,
                                                           float mask_value, const float* __restrict__ row_scale,
                                                           float* __restrict__ out, int64_t ldo,
                                                           float* __restrict__ sumsq_part, int nparts) {
  constexpr int d = DCH * 8;
  __shared__ uint32_t bitmap[kSimChunk];
  __shared__ float red[4][32];
  extern __shared__ __attribute__((aligned(16))) float stage_mem[];       // [2][32][kStagePitch]: 65 KB, dynamic
  float(*stage)[32][kStagePitch] = reinterpret_cast<float(*)[32][kStagePitch]>(stage_mem);
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
  // the factors of the 16 batch rows this lane's accumulator entries belong to
  float sc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t ub = u0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
    sc[r] = (row_scale && ub < B) ? row_scale[ub] : 1.f;
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
    float(*st)[kStagePitch] = stage[tile >> 3];
    const int col = (tile & 7) * kSimTile + n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * kh;
      const bool masked = (word >> m) & 1u;
      const float v = masked ? mask_value : acc[r] * sc[r];
      st[m][col] = j < I ? v : 0.f;                       // (columns past I: the zero padding of a pitched row)
      if (!masked && u0 + m < B && j < I) sq[r] = fmaf(v, v, sq[r]);
    }
  };
  // tiles wave, wave + 4 (group 0) and wave + 8, wave + 12 (group 1): the loads run one tile ahead
  if (wave < n_tiles) load_items(wave, ta);
  if (wave + 4 < n_tiles) load_items(wave + 4, tb);
  if (wave < n_tiles) do_tile(wave, ta);
  if (wave + 8 < n_tiles) load_items(wave + 8, ta);
  if (wave + 4 < n_tiles) do_tile(wave + 4, tb);
  if (wave + 12 < n_tiles) load_items(wave + 12, tb);
  // write-out of a group: wave w owns rows 8w .. 8w + 7, one row = one instruction per 256 items
  const bool vec_ok = ((ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  auto flush = [&](int g) {
    const int64_t cg = c0 + (int64_t)g * kSimGroup;
    if (cg >= ldo) return;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int m = 8 * wave + rr;
      if (u0 + m >= B) continue;
      float* __restrict__ orow = out + (u0 + m) * ldo + cg;
      if (vec_ok) {
        const int64_t c = cg + 4 * lane;
        const float4 v = *reinterpret_cast<const float4*>(&stage[g][m][4 * lane]);
        if (c + 3 < ldo) *reinterpret_cast<float4*>(orow + 4 * lane) = v;
        else {
          if (c < ldo) orow[4 * lane] = v.x;
          if (c + 1 < ldo) orow[4 * lane + 1] = v.y;
          if (c + 2 < ldo) orow[4 * lane + 2] = v.z;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (cg + lane + 64 * k < ldo) orow[lane + 64 * k] = stage[g][m][lane + 64 * k];
      }
    }
  };
  __syncthreads();                         // group 0 is staged (tiles 0 .. 7; absent tiles left their slots untouched:
  flush(0);                                // those columns are >= I rounded up to the tile, never inside ldo)
  if (wave + 8 < n_tiles) do_tile(wave + 8, ta);
  if (wave + 12 < n_tiles) do_tile(wave + 12, tb);
  __syncthreads();
  flush(1);
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

// ---------------------------------------------------------------------------------------------------------------
// Row norms of the masked score matrix WITHOUT forming it: |S_b|^2 = q_b^T (T^T T) q_b - sum over the row's masked
// items of (q_b . t_j)^2. The Gram matrix is d x d (one pass over the item table on the fp32 matrix pipe, block
// partials added in float64 and in block order by the last block to finish), the quadratic form is evaluated in
// float64 by one wave per batch row - so the tile kernel above can write the NORMALISED scores in its only pass
// (main.py:297 F.normalize(dim=1): x / max(|x|, eps)).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kGramTileFloats = 12288;      // 48 KB of item rows per block: 192 rows at d = 64
constexpr int kGramGroup = 8;               // blocks per first-level sum
__host__ __device__ constexpr int gram_rows(int d) { return kGramTileFloats / d; }

// Block b owns item rows [b * R, (b + 1) * R): staged in LDS with coalesced loads (one memory latency per block), then
// every wave accumulates output tiles G[32 ti .. , 32 tj ..] += rows^T rows on v_mfma_f32_32x32x2_f32 (two rows per
// instruction). The block partials are added in a FIXED order by whoever finishes last: per group of eight blocks into
// a float64 group sum, then the group sums into G - two levels so that no block sums more than a few dozen images.
template <int DT>                           // DT = d / 32 tiles per side
__global__ __launch_bounds__(kBlock) void usim_gram_kernel(const float* __restrict__ T, int64_t I, float* __restrict__ part,
                                                           double* __restrict__ part2, double* __restrict__ G,
                                                           uint32_t* __restrict__ ticket) {
  constexpr int d = DT * 32, R = gram_rows(d);
  __shared__ __attribute__((aligned(16))) float tile[kGramTileFloats];
  __shared__ int last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, kh = lane >> 5;
  const int64_t r0 = (int64_t)blockIdx.x * R;
  const int rows = (int)min((int64_t)R, I - r0);
  {
    const float4* src = reinterpret_cast<const float4*>(T + r0 * d);
    float4* dst = reinterpret_cast<float4*>(tile);
    const int n4 = rows * d / 4;
    for (int i = tid; i < R * d / 4; i += kBlock) dst[i] = i < n4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  float* __restrict__ mine = part + (size_t)blockIdx.x * d * d;
  for (int t = wave; t < DT * DT; t += 4) {             // output tile (ti, tj): G[32 ti + a][32 tj + b]
    const int ti = t / DT, tj = t % DT;
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
    for (int j = 0; j < R; j += 2)                       // (rows past `rows` are zeros)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(tile[(j + kh) * d + 32 * ti + n], tile[(j + kh) * d + 32 * tj + n], acc, 0,
                                                 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * kh;
      mine[(32 * ti + m) * d + 32 * tj + n] = acc[r];
    }
  }
  const int nb = (int)gridDim.x, ng = (nb + kGramGroup - 1) / kGramGroup, g = (int)blockIdx.x / kGramGroup;
  const int g_lo = g * kGramGroup, g_n = min(kGramGroup, nb - g_lo);
  __threadfence();
  __syncthreads();
  if (tid == 0) last = (atomicAdd(ticket + 1 + g, 1u) == (unsigned)g_n - 1) ? 1 : 0;
  __syncthreads();
  if (!last) return;
  __threadfence();
  for (int e = tid; e < d * d; e += kBlock) {
    double s = 0.0;
    for (int k = 0; k < g_n; ++k) s += (double)__builtin_nontemporal_load(part + (size_t)(g_lo + k) * d * d + e);
    part2[(size_t)g * d * d + e] = s;
  }
  __threadfence();
  __syncthreads();
  if (tid == 0) {
    ticket[1 + g] = 0u;
    last = (atomicAdd(ticket, 1u) == (unsigned)ng - 1) ? 1 : 0;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  for (int e = tid; e < d * d; e += kBlock) {
    double s = 0.0;
    for (int k = 0; k < ng; ++k) s += __builtin_nontemporal_load(part2 + (size_t)k * d * d + e);
    G[e] = s;
  }
  if (tid == 0) ticket[0] = 0u;
}

// one wave per batch row: inv[b] = 1 / max(sqrt(q^T G q - masked part), eps)
template <int DL>                           // DL = ceil(d / 64) columns per lane
__global__ __launch_bounds__(kBlock) void usim_norms_kernel(const float* __restrict__ Q, const int64_t* __restrict__ qidx,
                                                            int64_t B, const float* __restrict__ T, int d,
                                                            const double* __restrict__ G,
                                                            const int32_t* __restrict__ m_rowptr,
                                                            const void* __restrict__ m_cols, int m_stride, float eps,
                                                            float* __restrict__ inv_out) {
  __shared__ float qs[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t b = (int64_t)blockIdx.x * 4 + wave;
  if (b >= B) return;
  const int64_t r = qidx ? qidx[b] : b;
  const float* __restrict__ q = Q + r * d;
  for (int c = lane; c < d; c += 64) qs[wave][c] = q[c];
  __builtin_amdgcn_wave_barrier();
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < DL; ++k) {
    const int c = lane + 64 * k;
    if (c < d) {
      double y = 0.0;
      for (int a = 0; a < d; ++a) y += (double)qs[wave][a] * G[(size_t)a * d + c];
      s += y * (double)qs[wave][c];
    }
  }
  if (m_rowptr) {
    for (int e = m_rowptr[r]; e < m_rowptr[r + 1]; ++e) {
      const float* __restrict__ t = T + (int64_t)mask_col(m_cols, m_stride, e) * d;
      float p = 0.f;
#pragma unroll
      for (int k = 0; k < DL; ++k) {
        const int c = lane + 64 * k;
        if (c < d) p = fmaf(qs[wave][c], t[c], p);
      }
#pragma unroll
      for (int mk = 1; mk < 64; mk <<= 1) p += __shfl_xor(p, mk, kWave);
      if (lane == 0) s -= (double)p * (double)p;
    }
  }
#pragma unroll
  for (int mk = 1; mk < 64; mk <<= 1) s += __shfl_xor(s, mk, kWave);
  if (lane == 0) inv_out[b] = 1.f / fmaxf(sqrtf((float)fmax(s, 0.0)), eps);
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
constexpr int kTopkMaxCols = 36864;        // 144 keys per thread in registers
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

// One block per row, the row's keys in REGISTERS (CT per thread, column j = tid + 256 i: coalesced loads). Round 5 kept
// them in LDS and re-read all of them in each of the 32 bisection steps: 250 us per 1024 x 18 357 block, bound by the
// latency of un-pipelined LDS reads. From registers a step is CT compare-adds + one barrier.
template <int CT>
__global__ __launch_bounds__(kBlock) void topk_rows_kernel(const float* __restrict__ X, int64_t I, int64_t ldx, int K,
                                                           int64_t* __restrict__ idx_out, float* __restrict__ val_out) {
  __shared__ int red[2][4];
  __shared__ int slots;
  __shared__ uint32_t win_key[kTopkMaxK];
  __shared__ int32_t win_idx[kTopkMaxK];
  static_assert(kTopkMaxK == kBlock, "the block-wide sort gives every thread one winner slot");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.x;
  const float* __restrict__ row = X + b * ldx;
  uint32_t key[CT];
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    const int64_t j = tid + (int64_t)kBlock * i;
    key[i] = j < I ? order_key(row[j]) : 0u;               // 0 sorts below every real value (-inf included)
  }
  if (tid < kTopkMaxK) { win_key[tid] = 0u; win_idx[tid] = 0x7fffffff; }
  if (tid == 0) slots = 0;
  const int Ke = (int)min((int64_t)K, I);
  int phase = 0;
  auto block_count = [&](int c) {                          // every thread gets the block's total; one barrier per call
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) c += __shfl_xor(c, m, kWave);
    if (lane == 0) red[phase][wave] = c;
    __syncthreads();
    const int t = (red[phase][0] + red[phase][1]) + (red[phase][2] + red[phase][3]);
    phase ^= 1;
    return t;
  };
  // K-th largest key: the largest t with #{keys >= t} >= Ke, one bit at a time
  uint32_t tau = 0u;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t cand = tau | (1u << bit);
    int c = 0;
#pragma unroll
    for (int i = 0; i < CT; ++i) c += key[i] >= cand ? 1 : 0;
    if (block_count(c) >= Ke) tau = cand;
  }
  int gt = 0, eq = 0;
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    gt += key[i] > tau ? 1 : 0;
    eq += key[i] == tau ? 1 : 0;
  }
  const int n_gt = block_count(gt), n_eq = block_count(eq);
  const int need_eq = Ke - n_gt;                           // >= 1 entries equal to tau are winners: those of smallest id
  // the id below which an entry equal to tau is a winner (all of them unless the tie is cut: then the need_eq-th
  // smallest id among them, found by the same bisection over the 16-bit ids)
  int64_t id_cut = (int64_t)1 << 40;
  if (n_eq > need_eq) {
    uint32_t lim = 0u;                                     // largest L with #{eq entries with id < L} < need_eq ... as bits
    for (int bit = 16; bit >= 0; --bit) {
      const uint32_t cand = lim | (1u << bit);
      int c = 0;
#pragma unroll
      for (int i = 0; i < CT; ++i) c += (key[i] == tau && (uint32_t)(tid + kBlock * i) < cand) ? 1 : 0;
      if (block_count(c) < need_eq) lim = cand;            // fewer than need_eq ids below cand: cand is still too small
    }
    id_cut = (int64_t)lim + 1;                             // ids <= lim: exactly need_eq entries
  }
#pragma unroll
  for (int i = 0; i < CT; ++i) {
    const int64_t j = tid + (int64_t)kBlock * i;
    if (j < I && (key[i] > tau || (key[i] == tau && j < id_cut))) {
      const int p = atomicAdd(&slots, 1);                  // (the winners are sorted below: their slot order is free)
      if (p < kTopkMaxK) {
        win_key[p] = key[i];
        win_idx[p] = (int32_t)j;
      }
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


// ---------------------------------------------------------------------------------------------------------------
// evaluation metrics on the device (utility/batch_test.py:38-80 + utility/metrics.py of the reference): one thread per
// tested user looks its K_max ranked candidates up in the user's positives (sorted CSR row), forms precision / recall /
// ndcg / hit ratio @ every K in float64 and the block adds them up in thread order; the blocks' partial sums are added
// to the running totals in block order by the last block to finish. Nothing travels to the host per batch.
//   acc[4][n_ks] += sum over the batch's users;  order: precision, recall, ndcg, hit_ratio
// ---------------------------------------------------------------------------------------------------------------
constexpr int kEvalMaxKs = 8;
struct EvalKs {
  int k[kEvalMaxKs];
  int n;
};

__global__ __launch_bounds__(kBlock) void eval_metrics_kernel(const int32_t* __restrict__ rowptr,
                                                              const int32_t* __restrict__ cols,
                                                              const int64_t* __restrict__ rows, int64_t B, int K,
                                                              const int64_t* __restrict__ cand, EvalKs ks,
                                                              double* __restrict__ part, double* __restrict__ acc,
                                                              uint32_t* __restrict__ ticket) {
  __shared__ double red[4][4 * kEvalMaxKs];
  __shared__ int last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = (int64_t)blockIdx.x * kBlock + tid;
  double v[4 * kEvalMaxKs];
#pragma unroll
  for (int i = 0; i < 4 * kEvalMaxKs; ++i) v[i] = 0.0;
  if (b < B) {
    const int64_t r = rows[b];
    const int lo0 = rowptr[r], hi0 = rowptr[r + 1];
    const double n_pos = (double)(hi0 - lo0);
    int tot = 0;                                   // hits within K_max: the ideal list's ones (metrics.ndcg_at_k)
    int s[kEvalMaxKs];
    double dcg[kEvalMaxKs];
#pragma unroll
    for (int i = 0; i < kEvalMaxKs; ++i) { s[i] = 0; dcg[i] = 0.0; }
    for (int k = 0; k < K; ++k) {
      const int64_t c = cand[b * K + k];
      int lo = lo0, hi = hi0;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cols[mid] < c) lo = mid + 1;
        else hi = mid;
      }
      if (c >= 0 && lo < hi0 && cols[lo] == c) {
        ++tot;
        const double w = 1.0 / log2((double)(k + 2));
#pragma unroll
        for (int i = 0; i < kEvalMaxKs; ++i)
          if (i < ks.n && k < ks.k[i]) { ++s[i]; dcg[i] += w; }
      }
    }
#pragma unroll
    for (int i = 0; i < kEvalMaxKs; ++i) {
      if (i >= ks.n) continue;
      const int Kc = ks.k[i];
      double best = 0.0;
      for (int k = 0; k < min(tot, Kc); ++k) best += 1.0 / log2((double)(k + 2));
      v[0 * kEvalMaxKs + i] = (double)s[i] / (double)Kc;
      v[1 * kEvalMaxKs + i] = n_pos > 0.0 ? (double)s[i] / n_pos : 0.0;
      v[2 * kEvalMaxKs + i] = best > 0.0 ? dcg[i] / best : 0.0;
      v[3 * kEvalMaxKs + i] = s[i] > 0 ? 1.0 : 0.0;
    }
  }
  // thread order inside a wave (shuffle tree: fixed), wave order inside the block, block order across the launch
#pragma unroll
  for (int i = 0; i < 4 * kEvalMaxKs; ++i) {
    double x = v[i];
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) x += __shfl_xor(x, m, kWave);
    if (lane == 0) red[wave][i] = x;
  }
  __syncthreads();
  if (tid < 4 * kEvalMaxKs)
    part[(size_t)blockIdx.x * 4 * kEvalMaxKs + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
  __threadfence();
  __syncthreads();
  if (tid == 0) last = (atomicAdd(ticket, 1u) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (tid < 4 * kEvalMaxKs) {
    double x = acc[tid];
    for (unsigned k = 0; k < gridDim.x; ++k) x += __builtin_nontemporal_load(part + (size_t)k * 4 * kEvalMaxKs + tid);
    acc[tid] = x;
  }
  if (tid == 0) *ticket = 0u;
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
               const void* m_cols, int m_stride, float mask_value, const float* row_scale, float* out, int64_t ldo,
               float* sumsq_part, hipStream_t s) {
  const int nparts = (int)((I + kSimChunk - 1) / kSimChunk);
  const dim3 grid((unsigned)((B + 31) / 32), (unsigned)nparts);
  constexpr size_t stage_bytes = 2 * 32 * kStagePitch * sizeof(float);
#define SIM_CASE(DCH)                                                                                                   \
  {                                                                                                                     \
    static const int attr = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(sim_tiles_kernel<DCH>),              \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)stage_bytes);     \
    if (attr != 0) return MMSSL_E_UNSUPP;                                                                               \
    hipLaunchKernelGGL((sim_tiles_kernel<DCH>), grid, dim3(kBlock), stage_bytes, s, Q, qidx, B, T, I, m_rowptr, m_cols, \
                       m_stride, mask_value, row_scale, out, ldo, sumsq_part, nparts);                                   \
  }
  switch (d) {
    case 32: SIM_CASE(4); break;
    case 64: SIM_CASE(8); break;
    case 128: SIM_CASE(16); break;
    case 256:
      if (row_scale) return MMSSL_E_UNSUPP;        // (d = 256 keeps the partial-sums + scale-pass form)
      hipLaunchKernelGGL(sim_tiles_wide_kernel, grid, dim3(kBlock), 0, s, Q, qidx, B, T, I, m_rowptr, m_cols, m_stride,
                         mask_value, out, ldo, sumsq_part, nparts);
      break;
    default: return MMSSL_E_UNSUPP;
  }
#undef SIM_CASE
  MMSSL_LAUNCH_CHECK();
  return 0;
}

// inv[b] = 1 / max(|masked score row b|, eps) without forming the rows: Gram matrix of T + one wave per batch row.
// workspace: block partials (float), group sums and G (float64), tickets - usim_norms_workspace(d, n_items) bytes.
namespace {
struct GramWs {
  float* part;
  double *part2, *G;
  uint32_t* ticket;
  int nb, ng;
  size_t bytes;
};
inline GramWs gram_ws(void* ws, int64_t I, int d) {
  GramWs w;
  w.nb = (int)((I + gram_rows(d) - 1) / gram_rows(d));
  w.ng = (w.nb + kGramGroup - 1) / kGramGroup;
  char* p = reinterpret_cast<char*>(ws);
  w.ticket = reinterpret_cast<uint32_t*>(p);
  size_t o = ((size_t)(w.ng + 1) * 4 + 255) / 256 * 256;
  w.G = reinterpret_cast<double*>(p + o);
  o += (size_t)d * d * 8;
  w.part2 = reinterpret_cast<double*>(p + o);
  o += (size_t)w.ng * d * d * 8;
  w.part = reinterpret_cast<float*>(p + o);
  o += (size_t)w.nb * d * d * 4;
  w.bytes = o;
  return w;
}
}  // namespace
size_t usim_norms_workspace(int d, int64_t n_items) { return gram_ws(nullptr, n_items < 1 ? 1 : n_items, d).bytes + 256; }
int usim_norms_launch(const float* Q, const int64_t* qidx, int64_t B, const float* T, int64_t I, int d,
                      const int32_t* m_rowptr, const void* m_cols, int m_stride, float eps, float* inv_out, void* ws,
                      hipStream_t s) {
  if (d != 32 && d != 64 && d != 128) return MMSSL_E_UNSUPP;
  const GramWs w = gram_ws(ws, I, d);
  MMSSL_HIP_TRY(hipMemsetAsync(w.ticket, 0, (size_t)(w.ng + 1) * 4, s));
  float* part = w.part;
  double* G = w.G;
  switch (d) {
    case 32: hipLaunchKernelGGL((usim_gram_kernel<1>), dim3(w.nb), dim3(kBlock), 0, s, T, I, part, w.part2, G, w.ticket); break;
    case 64: hipLaunchKernelGGL((usim_gram_kernel<2>), dim3(w.nb), dim3(kBlock), 0, s, T, I, part, w.part2, G, w.ticket); break;
    default: hipLaunchKernelGGL((usim_gram_kernel<4>), dim3(w.nb), dim3(kBlock), 0, s, T, I, part, w.part2, G, w.ticket); break;
  }
  MMSSL_LAUNCH_CHECK();
  const dim3 grid((unsigned)((B + 3) / 4));
  if (d <= 64)
    hipLaunchKernelGGL((usim_norms_kernel<1>), grid, dim3(kBlock), 0, s, Q, qidx, B, T, d, G, m_rowptr, m_cols, m_stride,
                       eps, inv_out);
  else
    hipLaunchKernelGGL((usim_norms_kernel<2>), grid, dim3(kBlock), 0, s, Q, qidx, B, T, d, G, m_rowptr, m_cols, m_stride,
                       eps, inv_out);
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
  return sim_launch(Q, qidx, B, T, n_items, d, mask_rowptr, mask_cols, 4, mask_value, nullptr, out, ldo, sumsq_part,
                    as_stream(stream));
}

extern "C" size_t mmssl_usim_workspace_bytes(int d, int64_t n_items) {
  return (d == 32 || d == 64 || d == 128) ? usim_norms_workspace(d, n_items) : 0;
}

extern "C" int mmssl_usim_rows_f32(const float* Q, const int64_t* qidx, int64_t B, const float* T, int64_t n_items, int d,
                                   const int32_t* mask_rowptr, const int32_t* mask_cols, float eps, float* out, int64_t ldo,
                                   float* inv_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (B < 0 || n_items < 0 || ldo < n_items || !(eps > 0.f)) return MMSSL_E_BADARG;
  if (B == 0 || n_items == 0) return 0;
  if (!Q || !T || !out || !inv_out || (mask_rowptr && !mask_cols)) return MMSSL_E_BADARG;
  if (((uintptr_t)Q | (uintptr_t)T) & 15) return MMSSL_E_BADARG;
  if (mmssl_usim_workspace_bytes(d, n_items) == 0) return MMSSL_E_UNSUPP;
  if (!workspace || workspace_bytes < mmssl_usim_workspace_bytes(d, n_items)) return MMSSL_E_WORKSPACE;
  int rc = usim_norms_launch(Q, qidx, B, T, n_items, d, mask_rowptr, mask_cols, 4, eps, inv_out, workspace,
                             as_stream(stream));
  if (rc) return rc;
  return sim_launch(Q, qidx, B, T, n_items, d, mask_rowptr, mask_cols, 4, 0.f, inv_out, out, ldo, nullptr,
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
  // keys per thread: the smallest of 24 / 72 / 144 that covers the row (6 144 / 18 432 / 36 864 columns)
  const dim3 grid((unsigned)B), blk(kBlock);
  hipStream_t s = as_stream(stream);
  if (n_cols <= 24 * kBlock) hipLaunchKernelGGL((topk_rows_kernel<24>), grid, blk, 0, s, X, n_cols, ldx, K, idx_out, val_out);
  else if (n_cols <= 72 * kBlock) hipLaunchKernelGGL((topk_rows_kernel<72>), grid, blk, 0, s, X, n_cols, ldx, K, idx_out, val_out);
  else hipLaunchKernelGGL((topk_rows_kernel<144>), grid, blk, 0, s, X, n_cols, ldx, K, idx_out, val_out);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t mmssl_eval_workspace_bytes(int64_t B) {
  return (size_t)((B + kBlock - 1) / kBlock + 1) * 4 * kEvalMaxKs * sizeof(double) + 16;
}

extern "C" int mmssl_eval_accumulate_f64(const int32_t* pos_rowptr, const int32_t* pos_cols, const int64_t* rows, int64_t B,
                                         int K, const int64_t* cand, const int* ks, int n_ks, double* acc, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  if (B < 0 || K < 1 || n_ks < 1 || n_ks > kEvalMaxKs || !ks) return MMSSL_E_BADARG;
  if (B == 0) return 0;
  if (!pos_rowptr || !rows || !cand || !acc) return MMSSL_E_BADARG;
  if (!workspace || workspace_bytes < mmssl_eval_workspace_bytes(B)) return MMSSL_E_WORKSPACE;
  EvalKs e;
  e.n = n_ks;
  for (int i = 0; i < kEvalMaxKs; ++i) e.k[i] = i < n_ks ? ks[i] : 0;
  for (int i = 0; i < n_ks; ++i)
    if (ks[i] < 1 || ks[i] > K) return MMSSL_E_BADARG;
  const unsigned nb = (unsigned)((B + kBlock - 1) / kBlock);
  uint32_t* ticket = reinterpret_cast<uint32_t*>(workspace);
  double* part = reinterpret_cast<double*>(reinterpret_cast<char*>(workspace) + 16);
  MMSSL_HIP_TRY(hipMemsetAsync(ticket, 0, 4, as_stream(stream)));
  hipLaunchKernelGGL(eval_metrics_kernel, dim3(nb), dim3(kBlock), 0, as_stream(stream), pos_rowptr, pos_cols, rows, B, K, cand, e,
                     part, acc, ticket);
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
