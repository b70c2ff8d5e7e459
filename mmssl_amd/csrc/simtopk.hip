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

constexpr int kSimChunk = 256;           // items per block = one LDS stage: 72 x 32 short blocks for the Baby shape (no tail)
constexpr int kSimTile = 32;
constexpr int kStagePitch = kSimChunk + 4;

__device__ __forceinline__ int32_t mask_col(const void* cols, int stride, int64_t e) {
  return *reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(cols) + e * stride);
}

// DCH = d / 8: number of 8-deep k chunks (one float4 per lane half)
// A block = 32 batch rows x 256 items: every wave computes two 32 x 32 MFMA tiles (both tiles' item rows are requested
// from L2 before the first MFMA, straight into fragment layout), leaves them - scaled by the row's factor - in an LDS
// stage of 32 rows x 256 items, and the block then writes each staged row as ONE 1 KB segment (round 5 stored the MFMA
// fragments directly: 128-byte segments, 0.18 of the write roofline together with the separate scale pass).
// row_scale != NULL: out = row_scale[b] * score (the factor of F.normalize, known BEFORE this launch: usim_norms_kernel),
// so the [B, n_items] matrix is written exactly once. Columns [I, ldo) of a padded row are written as zeros.
// INK (mask IN the Kernel): the CSR mask as a per-block bitmap + the fixed-order partial sums of squares of the unmasked
// scores - the form the partial-sums API needs. The product's calls run INK = false: no bitmap, no binary searches in
// front of the tiles, and mask_scatter_kernel overwrites the (few) masked entries afterwards.
// decomposition builds for tools/frows_probe.py (never set in the product build): bit 0 = no MFMAs, bit 1 = no global
// stores, bit 2 = no item-row loads
#ifndef MMSSL_SIM_DBG
#define MMSSL_SIM_DBG 0
#endif
// A block walks `cpb` consecutive chunks with ITS 32 batch rows (gathered once, kept in registers with their factors): the
// fixed cost of a block - index load, row gather, factor loads: three dependent memory latencies - is paid once. Inside a
// chunk a wave loads one tile's item rows, waits, runs its 32 MFMAs and stages the tile - no software pipeline: at four
// waves per SIMD (<= 128 VGPRs) the waves of the resident blocks fall out of step after their first chunk (the matrix pipe
// serves one of them at a time) and one wave's loads and stores run under another's MFMAs (three waves per SIMD at d <= 64). Measured decomposition of the
// one-chunk-per-block form (profiles/r06/sim_tiles_ablate.txt): MFMAs 15 us + item-row loads 15 us (32 cache lines per
// load instruction) + stores 12 us + skeleton 10 us, all in series = 52 us.
template <int DCH, bool INK, bool VEC>
__global__ __launch_bounds__(kBlock, DCH <= 8 ? 3 : 2) void sim_tiles_kernel(const float* __restrict__ Q, const int64_t* __restrict__ qidx,
                                                              int64_t B, const float* __restrict__ T, int64_t I,
                                                              const int32_t* __restrict__ m_rowptr,
                                                              const void* __restrict__ m_cols, int m_stride,
                                                              float mask_value, const float* __restrict__ row_scale,
                                                              float* __restrict__ out, int64_t ldo,
                                                              float* __restrict__ sumsq_part, int nparts, int cpb) {
  constexpr int d = DCH * 8;
  __shared__ uint32_t bitmap[INK ? kSimChunk : 1];
  __shared__ float red[4][32];
  __shared__ __attribute__((aligned(16))) float stage[32][kStagePitch];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, kh = lane >> 5;
  const int64_t u0 = (int64_t)blockIdx.x * 32;
  const int ch_lo = (int)blockIdx.y * cpb, ch_hi = min(nparts, ch_lo + cpb);
  // A operand: my batch row (lane half kh holds k = 8q + 4kh .. +3 of every 8-chunk), zero past B
  const int64_t ub_q = u0 + n;
  const int64_t qrow_id = ub_q < B ? (qidx ? qidx[ub_q] : ub_q) : -1;
  float4 qf[DCH];
  {
    const float4* qrow = reinterpret_cast<const float4*>(Q + max(qrow_id, (int64_t)0) * d);
#pragma unroll
    for (int q = 0; q < DCH; ++q) {
      const float4 v = qrow[2 * q + kh];
      qf[q] = qrow_id >= 0 ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // the factors of the 16 batch rows this lane's accumulator entries belong to
  float sc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t ub = min(u0 + (r & 3) + 8 * (r >> 2) + 4 * kh, B - 1);
    sc[r] = row_scale ? row_scale[ub] : 1.f;
  }
  float sq[INK ? 16 : 1];
  auto tile = [&](int chunk, int t) {                      // one 32 x 32 tile of a chunk: load, MFMAs, stage
    const int64_t j = (int64_t)chunk * kSimChunk + (int64_t)t * kSimTile + n;
    const float4* trow = reinterpret_cast<const float4*>(T + ((MMSSL_SIM_DBG & 4) ? (int64_t)n : min(j, I - 1)) * d);
    float4 tf[DCH];
#pragma unroll
    for (int q = 0; q < DCH; ++q) tf[q] = trow[2 * q + kh];
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int q = 0; q < DCH; ++q) {
      if (MMSSL_SIM_DBG & 1) {
        acc[q & 15] += qf[q].x * tf[q].x + qf[q].y * tf[q].y + qf[q].z * tf[q].z + qf[q].w * tf[q].w;
        continue;
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[q].x, tf[q].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[q].y, tf[q].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[q].z, tf[q].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qf[q].w, tf[q].w, acc, 0, 0, 0);
    }
    const uint32_t word = INK ? bitmap[t * kSimTile + n] : 0u;
    const int col = t * kSimTile + n;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * kh;
      const bool masked = INK && ((word >> m) & 1u);
      const float v = masked ? mask_value : acc[r] * sc[r];
      stage[m][col] = j < I ? v : 0.f;                    // (columns past I: the zero padding of a pitched row)
      if (INK && !masked && u0 + m < B && j < I) sq[INK ? r : 0] = fmaf(v, v, sq[INK ? r : 0]);
    }
  };
  for (int chunk = ch_lo; chunk < ch_hi; ++chunk) {
    const int64_t c0 = (int64_t)chunk * kSimChunk;
    if (INK) {
      const int64_t c1 = min(I, c0 + kSimChunk);
#pragma unroll
      for (int r = 0; r < 16; ++r) sq[INK ? r : 0] = 0.f;
      for (int i = tid; i < kSimChunk; i += kBlock) bitmap[i] = 0u;
      __syncthreads();
      if (m_rowptr) {                      // thread (user ui, part): the user's masked items inside [c0, c1)
        const int ui = tid & 31, part = tid >> 5;
        if (u0 + ui < B) {
          const int64_t r = qidx ? qidx[u0 + ui] : (u0 + ui);
          int lo = m_rowptr[r], hi = m_rowptr[r + 1];
          const int end = hi;
          while (lo < hi) {                // lower bound of c0 in the sorted column list
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
      __syncthreads();
    }
    tile(chunk, wave);
    tile(chunk, wave + 4);
    __syncthreads();
    // wave w writes rows 8w .. 8w + 7, one row = one instruction; a row past B or a column past ldo is stored as a duplicate
    // of the last valid one (same data, read from the same clamped LDS slot): no branches around the stores
    if (!(MMSSL_SIM_DBG & 2) || stage[0][0] == 12345.f) {
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int m = (int)(min(u0 + 8 * wave + rr, B - 1) - u0);
        float* __restrict__ orow = out + (u0 + m) * ldo;
        if (VEC) {                         // host-checked: ldo % 4 == 0, out 16-byte aligned
          const int64_t c = min(c0 + 4 * lane, ldo - 4);
          *reinterpret_cast<float4*>(orow + c) = *reinterpret_cast<const float4*>(&stage[m][c - c0]);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int64_t c = min(c0 + lane + 64 * k, ldo - 1);
            orow[c] = stage[m][c - c0];
          }
        }
      }
    }
    if (INK && sumsq_part) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = sq[INK ? r : 0];
#pragma unroll
        for (int mk = 1; mk < 32; mk <<= 1) v += __shfl_xor(v, mk, kWave);
        if (n == 0) red[wave][(r & 3) + 8 * (r >> 2) + 4 * kh] = v;
      }
      __syncthreads();
      if (tid < 32 && u0 + tid < B)
        sumsq_part[(u0 + tid) * nparts + chunk] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
    }
    __syncthreads();                       // the stage (and `red`) are free for the next chunk
  }
}

// out[b, c] = value for every column c of CSR row qidx[b] of the mask: the masked entries of the score matrix, written
// after the tile kernel (one wave per batch row, its lanes over the row's entries: a 2 000-item user takes 32 steps)
__global__ __launch_bounds__(kBlock) void mask_scatter_kernel(const int64_t* __restrict__ qidx, int64_t B,
                                                              const int32_t* __restrict__ m_rowptr,
                                                              const void* __restrict__ m_cols, int m_stride, int64_t I,
                                                              float value, float* __restrict__ out, int64_t ldo) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int64_t r = qidx ? qidx[b] : b;
  const int lo = m_rowptr[r], hi = m_rowptr[r + 1];
  for (int e = lo + lane; e < hi; e += 64) {
    const int32_t c = mask_col(m_cols, m_stride, e);
    if (c >= 0 && c < I) out[b * ldo + c] = value;
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
__host__ __device__ constexpr int gram_rows(int d) { return kGramTileFloats / d; }

// Block b owns item rows [b * R, (b + 1) * R): staged in LDS with coalesced loads (one memory latency per block), then
// every wave accumulates output tiles G[32 ti .. , 32 tj ..] += rows^T rows on v_mfma_f32_32x32x2_f32 (two rows per
// instruction) and leaves them as the block's fp32 partial image.
template <int DT>                           // DT = d / 32 tiles per side
__global__ __launch_bounds__(kBlock) void usim_gram_kernel(const float* __restrict__ T, int64_t I, float* __restrict__ part) {
  constexpr int d = DT * 32, R = gram_rows(d);
  __shared__ __attribute__((aligned(16))) float tile[kGramTileFloats];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 31, kh = lane >> 5;
  const int64_t r0 = (int64_t)blockIdx.x * R;
  const int rows = (int)min((int64_t)R, I - r0);
  {
    const float4* src = reinterpret_cast<const float4*>(T + r0 * d);
    float4* dst = reinterpret_cast<float4*>(tile);
    const int n4 = rows * d / 4;
#pragma unroll 4
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
}

// G[e] = sum over the nb block images, in float64 and in a FIXED order: thread (e, p) adds images p, p + 4, ... (its loads
// issued eight at a time), the four partial sums of an element meet in LDS in order p = 0 .. 3. 64 elements per block.
__global__ __launch_bounds__(kBlock) void usim_gram_reduce_kernel(const float* __restrict__ part, int nb, int dd,
                                                                  double* __restrict__ G) {
  __shared__ double red[4][64];
  const int el = threadIdx.x & 63, p = threadIdx.x >> 6;
  const int e = (int)blockIdx.x * 64 + el;
  double s = 0.0;
  if (e < dd) {
    int k = p;
    for (; k + 28 < nb; k += 32) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + 4 * u) * dd + e];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (double)v[u];
    }
    for (; k < nb; k += 4) s += (double)part[(size_t)k * dd + e];
  }
  red[p][el] = s;
  __syncthreads();
  if (p == 0 && e < dd) G[e] = ((red[0][el] + red[1][el]) + red[2][el]) + red[3][el];
}

// one BLOCK per batch row: inv[b] = 1 / max(sqrt(q^T G q - sum over the row's masked items of (q . t_j)^2), eps), float64.
// Thread (c, p) forms the part of (G q)[c] over rows a = p, p + 4, ...; the masked items go one per thread, 256 at a time
// (a user with 2 000 train items takes 8 steps, not 2 000).
__global__ __launch_bounds__(kBlock) void usim_norms_kernel(const float* __restrict__ Q, const int64_t* __restrict__ qidx,
                                                            int64_t B, const float* __restrict__ T, int d,
                                                            const double* __restrict__ G,
                                                            const int32_t* __restrict__ m_rowptr,
                                                            const void* __restrict__ m_cols, int m_stride, float eps,
                                                            float* __restrict__ inv_out) {
  __shared__ float qs[128];
  __shared__ double red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.x;
  const int64_t r = qidx ? qidx[b] : b;
  const float* __restrict__ q = Q + r * d;
  if (tid < d) qs[tid] = q[tid];
  __syncthreads();
  double s = 0.0;
  // q^T G q = sum_c q[c] * sum_a q[a] G[a][c]: thread (c = tid % d, p = tid / d) takes rows a = p, p + 256 / d, ...
  {
    const int c = tid % d, p = tid / d, step = kBlock / d;             // d in {32, 64, 128}: step 8 / 4 / 2
    double y = 0.0;
    for (int a = p; a < d; a += step) y += (double)qs[a] * G[(size_t)a * d + c];
    s = y * (double)qs[c];
  }
  if (m_rowptr) {
    const int lo = m_rowptr[r], hi = m_rowptr[r + 1];
    for (int e = lo + tid; e < hi; e += kBlock) {
      const float4* __restrict__ t = reinterpret_cast<const float4*>(T + (int64_t)mask_col(m_cols, m_stride, e) * d);
      float pa = 0.f, pb = 0.f;
      for (int k = 0; k < d / 4; k += 2) {
        const float4 x = t[k], y2 = t[k + 1];
        pa = fmaf(qs[4 * k], x.x, fmaf(qs[4 * k + 1], x.y, fmaf(qs[4 * k + 2], x.z, fmaf(qs[4 * k + 3], x.w, pa))));
        pb = fmaf(qs[4 * k + 4], y2.x, fmaf(qs[4 * k + 5], y2.y, fmaf(qs[4 * k + 6], y2.z, fmaf(qs[4 * k + 7], y2.w, pb))));
      }
      const double pp = (double)(pa + pb);
      s -= pp * pp;
    }
  }
#pragma unroll
  for (int mk = 1; mk < 64; mk <<= 1) s += __shfl_xor(s, mk, kWave);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  if (tid == 0) {
    const double tot = (red[0] + red[1]) + (red[2] + red[3]);
    inv_out[b] = 1.f / fmaxf(sqrtf((float)fmax(tot, 0.0)), eps);
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
  // every load unconditional (a clamped address, the padding decided afterwards): a load under `if (j < I)` becomes a
  // branch of its own with a full wait behind it - 72 memory latencies one after the other
  float raw[CT];
#pragma unroll
  for (int i = 0; i < CT; ++i) raw[i] = row[min((int64_t)tid + (int64_t)kBlock * i, I - 1)];
#pragma unroll
  for (int i = 0; i < CT; ++i)
    key[i] = (tid + (int64_t)kBlock * i) < I ? order_key(raw[i]) : 0u;      // 0 sorts below every real value (-inf included)
  if (tid < kTopkMaxK) { win_key[tid] = 0u; win_idx[tid] = 0x7fffffff; }
  if (tid == 0) slots = 0;
  const int Ke = (int)min((int64_t)K, I);
  int phase = 0;
  // block total of per-WAVE counts (every lane of a wave passes the same value): four LDS words and one barrier
  auto block_total = [&](int c) {
    if (lane == 0) red[phase][wave] = c;
    __syncthreads();
    const int t = (red[phase][0] + red[phase][1]) + (red[phase][2] + red[phase][3]);
    phase ^= 1;
    return t;
  };
  // Candidates first. The Ke-th largest of the 256 THREAD MAXIMA is a lower bound of the row's Ke-th largest key (Ke
  // different entries are at least that large), so every winner is >= t0 - and for scores without long runs of ties only
  // a few more than Ke entries are (about Ke (1 + CT Ke / I)). Bisection over ONE value per thread is 32 x (compare,
  // popcount, barrier); the full bisection below costs CT compares per step and is only run when more than 256 entries
  // reach t0 (ties: rows that are mostly -inf, constant rows).
  uint32_t kmax = 0u;
#pragma unroll
  for (int i = 0; i < CT; ++i) kmax = max(kmax, key[i]);
  uint32_t t0 = 0u;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t cand = t0 | (1u << bit);
    if (block_total(__builtin_popcountll(__ballot(kmax >= cand))) >= Ke) t0 = cand;
  }
  int mine = 0;
#pragma unroll
  for (int i = 0; i < CT; ++i) mine += key[i] >= t0 ? 1 : 0;
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) mine += __shfl_xor(mine, m, kWave);
  const int n_cand = block_total(mine);
  int n_win = Ke;                          // entries of win_key / win_idx to sort; the first Ke of the sorted list are the answer
  if (n_cand <= kTopkMaxK) {
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int64_t j = tid + (int64_t)kBlock * i;
      if (j < I && key[i] >= t0) {
        const int p = atomicAdd(&slots, 1);
        win_key[p] = key[i];
        win_idx[p] = (int32_t)j;
      }
    }
    n_win = n_cand;
  } else {
    // K-th largest key: the largest t with #{keys >= t} >= Ke, one bit at a time
    uint32_t tau = 0u;
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t cand = tau | (1u << bit);
      int c = 0;
#pragma unroll
      for (int i = 0; i < CT; ++i) c += key[i] >= cand ? 1 : 0;
#pragma unroll
      for (int m = 1; m < 64; m <<= 1) c += __shfl_xor(c, m, kWave);
      if (block_total(c) >= Ke) tau = cand;
    }
    int gt = 0, eq = 0;
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      gt += key[i] > tau ? 1 : 0;
      eq += key[i] == tau ? 1 : 0;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
      gt += __shfl_xor(gt, m, kWave);
      eq += __shfl_xor(eq, m, kWave);
    }
    const int n_gt = block_total(gt), n_eq = block_total(eq);
    const int need_eq = Ke - n_gt;                         // >= 1 entries equal to tau are winners: those of smallest id
    // the id below which an entry equal to tau is a winner (all of them unless the tie is cut: then the need_eq-th
    // smallest id among them, found by the same bisection over the ids)
    int64_t id_cut = (int64_t)1 << 40;
    if (n_eq > need_eq) {
      uint32_t lim = 0u;
      for (int bit = 16; bit >= 0; --bit) {
        const uint32_t cand = lim | (1u << bit);
        int c = 0;
#pragma unroll
        for (int i = 0; i < CT; ++i) c += (key[i] == tau && (uint32_t)(tid + kBlock * i) < cand) ? 1 : 0;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) c += __shfl_xor(c, m, kWave);
        if (block_total(c) < need_eq) lim = cand;          // fewer than need_eq ids below cand: cand is still too small
      }
      id_cut = (int64_t)lim + 1;                           // ids <= lim: exactly need_eq entries
    }
#pragma unroll
    for (int i = 0; i < CT; ++i) {
      const int64_t j = tid + (int64_t)kBlock * i;
      if (j < I && (key[i] > tau || (key[i] == tau && j < id_cut))) {
        const int p = atomicAdd(&slots, 1);                // (the winners are sorted below: their slot order is free)
        if (p < kTopkMaxK) {
          win_key[p] = key[i];
          win_idx[p] = (int32_t)j;
        }
      }
    }
  }
  __syncthreads();
  if (n_win > 64) {                         // block-uniform: up to 256 candidates, bitonic network through LDS
    uint32_t k = tid < n_win ? win_key[tid] : 0u;
    int32_t id = tid < n_win ? win_idx[tid] : 0x7fffffff;
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
    const bool valid = tid < n_win;
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
    }
  }
  if (tid >= Ke && tid < K) {               // a row shorter than K: padding
    idx_out[b * K + tid] = -1;
    if (val_out) val_out[b * K + tid] = 0.f;
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
  // one WAVE per user: lane k looks candidate k (and k + 64, ...) up in the user's positives - 64 bisections side by
  // side instead of K one after the other - and the per-K sums come from popcounts / in-order lane sums of the hit mask
  __shared__ double red[4][4 * kEvalMaxKs];
  __shared__ double disc[kTopkMaxK + 1], cum[kTopkMaxK + 1];
  __shared__ int last;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k <= K; k += kBlock) disc[k] = 1.0 / log2((double)(k + 2));
  __syncthreads();
  for (int k = tid; k <= K; k += kBlock) {               // cum[t] = ideal DCG of t hits = disc[0] + ... + disc[t - 1], in order
    double c = 0.0;
    for (int j = 0; j < k; ++j) c += disc[j];
    cum[k] = c;
  }
  __syncthreads();
  double v[4 * kEvalMaxKs];
#pragma unroll
  for (int i = 0; i < 4 * kEvalMaxKs; ++i) v[i] = 0.0;
  // users of this wave: b = (4 blockIdx + wave) + 4 gridDim * t, in that order
  for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < B; b += (int64_t)gridDim.x * 4) {
    const int64_t r = rows[b];
    const int lo0 = rowptr[r], hi0 = rowptr[r + 1];
    const double n_pos = (double)(hi0 - lo0);
    int tot = 0, s[kEvalMaxKs];
    double dcg[kEvalMaxKs];
#pragma unroll
    for (int i = 0; i < kEvalMaxKs; ++i) { s[i] = 0; dcg[i] = 0.0; }
    for (int k0 = 0; k0 < K; k0 += 64) {
      const int k = k0 + lane;
      bool hit = false;
      if (k < K) {
        const int64_t c = cand[b * K + k];
        int lo = lo0, hi = hi0;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (cols[mid] < c) lo = mid + 1;
          else hi = mid;
        }
        hit = c >= 0 && lo < hi0 && cols[lo] == c;
      }
      const uint64_t mask = __ballot(hit);
      tot += __builtin_popcountll(mask);
#pragma unroll
      for (int i = 0; i < kEvalMaxKs; ++i) {
        if (i >= ks.n || ks.k[i] <= k0) continue;
        const int upto = ks.k[i] - k0;                                 // ranks k0 .. k0 + upto - 1 count for this K
        uint64_t m = upto >= 64 ? mask : (mask & ((1ull << upto) - 1ull));
        s[i] += __builtin_popcountll(m);
        while (m) {                                                    // ascending rank: a fixed order
          const int j = __builtin_ctzll(m);
          dcg[i] += disc[k0 + j];
          m &= m - 1;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kEvalMaxKs; ++i) {
      if (i >= ks.n) continue;
      const int Kc = ks.k[i];
      const double best = cum[min(tot, Kc)];
      v[0 * kEvalMaxKs + i] += (double)s[i] / (double)Kc;
      v[1 * kEvalMaxKs + i] += n_pos > 0.0 ? (double)s[i] / n_pos : 0.0;
      v[2 * kEvalMaxKs + i] += best > 0.0 ? dcg[i] / best : 0.0;
      v[3 * kEvalMaxKs + i] += s[i] > 0 ? 1.0 : 0.0;
    }
  }
  // every lane of a wave holds the same sums; wave order inside the block, block order across the launch
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < 4 * kEvalMaxKs; ++i) red[wave][i] = v[i];
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
  {   // the blocks' images in block order: thread (i, p) adds images p, p + 8, ..., the eight partial sums meet in order
    __shared__ double fin[8][4 * kEvalMaxKs];
    const int i = tid & 31, p = tid >> 5;
    double x = 0.0;
    for (unsigned k = p; k < gridDim.x; k += 8) x += __builtin_nontemporal_load(part + (size_t)k * 4 * kEvalMaxKs + i);
    fin[p][i] = x;
    __syncthreads();
    if (tid < 4 * kEvalMaxKs) {
      double t = acc[tid];
#pragma unroll
      for (int q = 0; q < 8; ++q) t += fin[q][tid];
      acc[tid] = t;
    }
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
  // chunks per block: about three resident blocks per CU in ONE round (768 on this part), each amortising its row gather
  const int64_t ub = (B + 31) / 32;
  int cpb = (int)((ub * nparts + 767) / 768);
  cpb = cpb < 1 ? 1 : (cpb > nparts ? nparts : cpb);
  const dim3 grid((unsigned)ub, (unsigned)((nparts + cpb - 1) / cpb));
  const dim3 grid_wide((unsigned)ub, (unsigned)nparts);
  // the partial-sums API needs the mask inside the tile kernel (sums over the UNMASKED scores); every other call runs the
  // bitmap-free kernel and overwrites the masked entries afterwards
  const bool ink = sumsq_part != nullptr && m_rowptr != nullptr;
  const bool vec = ((ldo & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && ldo >= 4;
#define SIM_LAUNCH(DCH, INK, VEC)                                                                                        \
  hipLaunchKernelGGL((sim_tiles_kernel<DCH, INK, VEC>), grid, dim3(kBlock), 0, s, Q, qidx, B, T, I, m_rowptr, m_cols,    \
                     m_stride, mask_value, row_scale, out, ldo, sumsq_part, nparts, cpb)
#define SIM_CASE(DCH)                                                                                                    \
  if (ink || sumsq_part) {                                                                                               \
    if (vec) SIM_LAUNCH(DCH, true, true);                                                                                \
    else SIM_LAUNCH(DCH, true, false);                                                                                   \
  } else {                                                                                                               \
    if (vec) SIM_LAUNCH(DCH, false, true);                                                                               \
    else SIM_LAUNCH(DCH, false, false);                                                                                  \
  }
  switch (d) {
    case 32: SIM_CASE(4); break;
    case 64: SIM_CASE(8); break;
    case 128: SIM_CASE(16); break;
    case 256:
      if (row_scale) return MMSSL_E_UNSUPP;        // (d = 256 keeps the partial-sums + scale-pass form)
      hipLaunchKernelGGL(sim_tiles_wide_kernel, grid_wide, dim3(kBlock), 0, s, Q, qidx, B, T, I, m_rowptr, m_cols, m_stride,
                         mask_value, out, ldo, sumsq_part, nparts);
      break;
    default: return MMSSL_E_UNSUPP;
  }
#undef SIM_CASE
#undef SIM_LAUNCH
  MMSSL_LAUNCH_CHECK();
  if (d != 256 && !sumsq_part && m_rowptr) {
    hipLaunchKernelGGL(mask_scatter_kernel, dim3((unsigned)((B + 3) / 4)), dim3(kBlock), 0, s, qidx, B, m_rowptr, m_cols,
                       m_stride, I, mask_value, out, ldo);
    MMSSL_LAUNCH_CHECK();
  }
  return 0;
}

// inv[b] = 1 / max(|masked score row b|, eps) without forming the rows: Gram matrix of T + one wave per batch row.
// workspace: block partials (float), group sums and G (float64), tickets - usim_norms_workspace(d, n_items) bytes.
namespace {
struct GramWs {
  float* part;
  double* G;
  int nb;
  size_t bytes;
};
inline GramWs gram_ws(void* ws, int64_t I, int d) {
  GramWs w;
  w.nb = (int)((I + gram_rows(d) - 1) / gram_rows(d));
  char* p = reinterpret_cast<char*>(ws);
  w.G = reinterpret_cast<double*>(p);
  size_t o = (size_t)d * d * 8;
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
  switch (d) {
    case 32: hipLaunchKernelGGL((usim_gram_kernel<1>), dim3(w.nb), dim3(kBlock), 0, s, T, I, w.part); break;
    case 64: hipLaunchKernelGGL((usim_gram_kernel<2>), dim3(w.nb), dim3(kBlock), 0, s, T, I, w.part); break;
    default: hipLaunchKernelGGL((usim_gram_kernel<4>), dim3(w.nb), dim3(kBlock), 0, s, T, I, w.part); break;
  }
  MMSSL_LAUNCH_CHECK();
  hipLaunchKernelGGL(usim_gram_reduce_kernel, dim3((unsigned)((d * d + 63) / 64)), dim3(kBlock), 0, s, w.part, w.nb, d * d,
                     w.G);
  MMSSL_LAUNCH_CHECK();
  hipLaunchKernelGGL(usim_norms_kernel, dim3((unsigned)B), dim3(kBlock), 0, s, Q, qidx, B, T, d, w.G, m_rowptr, m_cols,
                     m_stride, eps, inv_out);
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

namespace {
inline unsigned eval_blocks(int64_t B) {          // one wave per user, at most 512 blocks (their images are summed by one)
  const int64_t nb = (B + 3) / 4;
  return (unsigned)(nb < 1 ? 1 : (nb > 512 ? 512 : nb));
}
}  // namespace
extern "C" size_t mmssl_eval_workspace_bytes(int64_t B) {
  return (size_t)(eval_blocks(B) + 1) * 4 * kEvalMaxKs * sizeof(double) + 16;
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
  if (K > kTopkMaxK) return MMSSL_E_UNSUPP;
  for (int i = 0; i < n_ks; ++i)
    if (ks[i] < 1 || ks[i] > K) return MMSSL_E_BADARG;
  const unsigned nb = eval_blocks(B);
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
