// Per-modality projection GEMM on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32 / 16x16x4: exact fp32).
// nn.Linear image_trans / text_trans + nn.Dropout and their autograd, one modality per call
// (/root/reference/MMSSL/Models.py:28-29, 54, 173-174):
//
//   forward : Y[M,N]  = dropout(F[M,K] . W[N,K]^T + b)          (F streamed once from HBM)
//   wgrad   : gW[N,K] = gY[M,N]^T . F[M,K],  gb[N] = colsum(gY)
//
// The single-GPU hot path runs the GROUPED kernels of csrc/projection.hip (all modalities in one launch); these
// per-modality kernels serve the row-sharded step (mmssl_amd/dist.py), modality lists the grouped kernels do not take,
// and the discriminator-side products of ops.sim_rows_bwd. One kernel per job:
//   gemm_sk_kernel   forward, K % 32 == 0: stream-K over (64x64 tile, 32-deep slice) units, 4-stage LDS-DMA ring,
//                    double-buffered fragments; sk_reduce_kernel adds partial slots in block order (+ bias, dropout)
//   wgrad10_kernel   weight gradient, K % 64 == 0 and N % 64 == 0: register-direct 16x16x4 MFMA fragments straight
//                    from global memory, dropout backward and bias gradient on the loaded fragment
//   gemm64_kernel    every other shape (K % 4 == 0): register-staged 64x64 tiles, split-K with a fixed-order reduce
// (The round-1/round-2 experiment generations - LDS-DMA split-K, ping-pong, eight-wave, register-A, transposed-feature
//  and split-bf16 products - were measured, documented in DESIGN.md section 4 and removed from the build.)
#include <cstdlib>
#include <type_traits>

#include "lds_dma.hpp"

using namespace mmssl;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BT = 64;    // output tile edge
constexpr int BK = 32;    // reduction slice
// LDS row stride (floats) of gemm64_kernel's k-major operand image: 65 when the slice is TRANSPOSED into it
// (conflict-free ds_write_b32 and fragment ds_read_b32); 68 when it is copied as is (wgrad): rows stay 16-B
// aligned, so a thread's float4 is ONE conflict-free ds_write_b128 instead of four 2-way-conflicting
// ds_write_b32 (PMC: 25 % of the wgrad kernel's LDS cycles were bank conflicts with stride 65).
template <bool DIRECT>
constexpr int ld_of() { return DIRECT ? 68 : 65; }

// DIRECT = false: operands are row-major [i][kk] (forward: F[M,K], W[N,K]) -> transposed into LDS
// DIRECT = true : operands are row-major [kk][i] (wgrad: gY[M,N], F[M,K])  -> copied as is
// `mk` (DIRECT only, may be NULL): uint8 keep-mask with the operand's layout; kept entries are scaled
// by `ms`, dropped ones zeroed — the dropout backward applied to gY while it is fetched for wgrad.
template <bool DIRECT>
__device__ __forceinline__ void fetch_slice(const float* __restrict__ P, int64_t ld, int64_t i0, int64_t I,
                                            int64_t kk0, int64_t kk_end, float4 (&r)[2],
                                            const uint8_t* __restrict__ mk = nullptr, float ms = 1.f) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    r[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!DIRECT) {
      const int64_t i = i0 + (tid >> 3) + 32 * p;
      const int64_t kk = kk0 + 4 * (tid & 7);
      if (i < I && kk < kk_end) r[p] = *reinterpret_cast<const float4*>(P + i * ld + kk);
    } else {
      const int64_t kk = kk0 + (tid >> 4) + 16 * p;
      const int64_t i = i0 + 4 * (tid & 15);
      if (kk < kk_end && i < I) {
        r[p] = *reinterpret_cast<const float4*>(P + kk * ld + i);
        if (mk) {
          const uchar4 k = *reinterpret_cast<const uchar4*>(mk + kk * ld + i);
          r[p].x = k.x ? r[p].x * ms : 0.f;
          r[p].y = k.y ? r[p].y * ms : 0.f;
          r[p].z = k.z ? r[p].z * ms : 0.f;
          r[p].w = k.w ? r[p].w * ms : 0.f;
        }
      }
    }
  }
}

template <bool DIRECT>
__device__ __forceinline__ void store_slice(float* __restrict__ S, const float4 (&r)[2]) {
  constexpr int LD = ld_of<DIRECT>();
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    if (!DIRECT) {
      const int i = (tid >> 3) + 32 * p, k = 4 * (tid & 7);
      S[(k + 0) * LD + i] = r[p].x;
      S[(k + 1) * LD + i] = r[p].y;
      S[(k + 2) * LD + i] = r[p].z;
      S[(k + 3) * LD + i] = r[p].w;
    } else {
      const int k = (tid >> 4) + 16 * p, i = 4 * (tid & 15);
      *reinterpret_cast<float4*>(S + k * LD + i) = r[p];
    }
  }
}

// C[split][i][j] (+)= sum_{kk in split} A(i,kk) * B(j,kk)
template <bool DIRECT>
__global__ __launch_bounds__(kBlock) void gemm64_kernel(const float* __restrict__ A, int64_t lda,
                                                        const float* __restrict__ B, int64_t ldb, int64_t I,
                                                        int64_t J, int64_t KK, int64_t kk_chunk,
                                                        float* __restrict__ C, int64_t ldc,
                                                        int64_t split_stride, const float* __restrict__ bias,
                                                        const uint8_t* __restrict__ keep, float scale,
                                                        const uint8_t* __restrict__ maskA, float scaleA) {
  constexpr int LD = ld_of<DIRECT>();
  __shared__ __attribute__((aligned(16))) float As[2][BK * LD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t i0 = (int64_t)blockIdx.x * BT, j0 = (int64_t)blockIdx.y * BT;
  const int64_t kk_beg = (int64_t)blockIdx.z * kk_chunk;
  const int64_t kk_end = min(KK, kk_beg + kk_chunk);
  const int nk = (int)((kk_end - kk_beg + BK - 1) / BK);
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 ra[2], rb[2];
  if (nk > 0) {
    fetch_slice<DIRECT>(A, lda, i0, I, kk_beg, kk_end, ra, maskA, scaleA);
    fetch_slice<DIRECT>(B, ldb, j0, J, kk_beg, kk_end, rb);
    store_slice<DIRECT>(As[0], ra);
    store_slice<DIRECT>(Bs[0], rb);
  }
  __syncthreads();
  const int frag = (lane >> 5) * LD + (lane & 31);
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {   // next slice: global -> registers while this slice computes
      fetch_slice<DIRECT>(A, lda, i0, I, kk_beg + (int64_t)(kt + 1) * BK, kk_end, ra, maskA, scaleA);
      fetch_slice<DIRECT>(B, ldb, j0, J, kk_beg + (int64_t)(kt + 1) * BK, kk_end, rb);
    }
    const float* as = As[buf] + frag + wm * 32;
    const float* bs = Bs[buf] + frag + wn * 32;
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      const float a = as[2 * s * LD];
      const float b = bs[2 * s * LD];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (kt + 1 < nk) {
      store_slice<DIRECT>(As[buf ^ 1], ra);
      store_slice<DIRECT>(Bs[buf ^ 1], rb);
    }
    __syncthreads();
  }
  // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int64_t col = j0 + wn * 32 + (lane & 31);
  float* Cp = C + (int64_t)blockIdx.z * split_stride;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t row = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < I && col < J) {
      float v = acc[r];
      if (bias) v += bias[col];
      if (keep) v = keep[row * J + col] ? v * scale : 0.f;
      Cp[row * ldc + col] = v;
    }
  }
}



constexpr int kDmaStages = 4;
constexpr int kDmaStageFloats = 2 * BT * BK;        // A slice then B slice: 8 KB + 8 KB

// ======================================================================================
// Stream-K forward kernel: 64x64 block tile, 2x2 waves (one 32x32 MFMA accumulator each), 32-deep slices, a 4-stage
// LDS ring filled by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass; the XOR swizzle
// chunk c of row i -> slot i*8 + (c ^ ((i >> 1) & 7)) is applied to each lane's SOURCE address because LDS-DMA writes
// lane-linearly), fragments back as ds_read_b128 (a lane's float4 = 4 consecutive k feeds 4 MFMA steps).
//
//  * WORK DECOMPOSITION. The (tile, slice) units are laid out on one axis and cut into as many EQUAL contiguous
//    ranges as there are block slots (Baby image projection: 287 x 128 = 36736 units / 512 = 71.75 per block; a
//    tiles x splits grid of equal blocks would run 2.24 rounds there, the last on a quarter of the chip). A range
//    that covers a whole tile stores the result directly (bias + dropout in the epilogue); partial ranges store
//    the raw accumulator to one of the block's two slots (head / tail) and sk_reduce_kernel adds a tile's slots in
//    block order (fixed order: deterministic). Partial traffic: <= 2 x 16 KB per block.
//  * FRAGMENT DOUBLE BUFFERING. The fragments of slice kt+1 are fetched into a second register set BEFORE the MFMAs
//    of slice kt are issued (reading them right after the barrier put barrier + LDS latency in front of every
//    slice's MFMA chain: SQ_WAIT_INST_ANY 66 % of wave cycles, MFMA pipe 59 % busy in the round-1 profile).
// Preconditions (host-checked): KK % 32 == 0, row-major [i][kk] operands, 16-B aligned rows.
// ======================================================================================
constexpr int kSkTileFloats = BT * BT;      // one partial slot: the block's 64x64 accumulator image (16 KB)

struct Frag {
  float4 a[4], b[4];
};

__global__ __launch_bounds__(kBlock) void gemm_sk_kernel(const float* __restrict__ A, int64_t lda,
                                                         const float* __restrict__ B, int64_t ldb, int64_t I,
                                                         int64_t J, int tiles_j, int S, int64_t total_units, int upb,
                                                         float* __restrict__ C, int64_t ldc,
                                                         const float* __restrict__ bias,
                                                         const uint8_t* __restrict__ keep, float scale,
                                                         float* __restrict__ partials) {
  __shared__ __attribute__((aligned(16))) float ring[kDmaStages * kDmaStageFloats];     // 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, lr = lane & 31;
  const int sw = (lr >> 1) & 7;
  const int ia = (wm * 32 + lr) * BK, jb = BT * BK + (wn * 32 + lr) * BK;
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  const unsigned piece = __builtin_amdgcn_readfirstlane((unsigned)(16 * wave * BK * 4));
  const int64_t u_begin = (int64_t)blockIdx.x * upb;
  const int64_t u_end = min(total_units, u_begin + upb);
  int64_t u = u_begin;
  while (u < u_end) {
    const int64_t tile = u / S;
    const int s0 = (int)(u - tile * S);
    const int s1 = (int)min((int64_t)S, s0 + (u_end - u));
    const int nk = s1 - s0;
    const int64_t i0 = (tile / tiles_j) * BT, j0 = (tile % tiles_j) * BT;
    // DMA pieces: wave w moves rows [16w, 16w+16) of each operand slice as 2 x (8 rows x 128 B)
    const float* pa[2];
    const float* pb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 16 * wave + 8 * j + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);                  // source-side swizzle
      pa[j] = A + min(i0 + r, I - 1) * lda + (int64_t)s0 * BK + 4 * c;
      pb[j] = B + min(j0 + r, J - 1) * ldb + (int64_t)s0 * BK + 4 * c;
    }
    auto issue = [&](int kt) {
      const unsigned st = ring_lds + (unsigned)(kt & (kDmaStages - 1)) * (kDmaStageFloats * 4) + piece;
      glds16(pa[0] + (int64_t)kt * BK, st);
      glds16(pa[1] + (int64_t)kt * BK, st + 8 * BK * 4);
      glds16(pb[0] + (int64_t)kt * BK, st + BT * BK * 4);
      glds16(pb[1] + (int64_t)kt * BK, st + BT * BK * 4 + 8 * BK * 4);
    };
    auto read_frags = [&](int kt, Frag& f) {
      const float* st = ring + (kt & (kDmaStages - 1)) * kDmaStageFloats;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int pos = ((2 * q + h) ^ sw) * 4;
        f.a[q] = *reinterpret_cast<const float4*>(st + ia + pos);
        f.b[q] = *reinterpret_cast<const float4*>(st + jb + pos);
      }
    };
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto mfma16 = [&](const Frag& f) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[q].x, f.b[q].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[q].y, f.b[q].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[q].z, f.b[q].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[q].w, f.b[q].w, acc, 0, 0, 0);
      }
    };
    // one pipeline step: make slice kt+1 resident, fetch its fragments into `nxt`, then run slice kt from `cur`
    auto step = [&](int kt, const Frag& cur, Frag& nxt) {
      if (kt + 1 < nk) {
        const int after = min(nk - 1, kt + 3) - (kt + 1);     // DMA groups issued after slice kt+1
        if (after >= 2) vm_wait_n<8>();
        else if (after == 1) vm_wait_n<4>();
        else vm_wait_n<0>();
        lgkm_wait0();        // my fragment reads of slice kt are done: its stage may be refilled after the barrier
        bare_barrier();      // everyone's pieces of slice kt+1 have landed; nobody reads stage kt & 3 any more
        if (kt + 4 < nk) issue(kt + 4);
        read_frags(kt + 1, nxt);
      }
      mfma16(cur);
    };
    // the previous segment's epilogue stores and fragment reads must be done before the ring is refilled
    vm_wait_n<0>();
    lgkm_wait0();
    bare_barrier();
    issue(0);
    if (nk > 1) issue(1);
    if (nk > 2) issue(2);
    if (nk > 2) vm_wait_n<8>();
    else if (nk > 1) vm_wait_n<4>();
    else vm_wait_n<0>();
    bare_barrier();
    if (nk > 3) issue(3);
    // steady state (slices kt+1 .. kt+4 exist): branch-free, so the accumulator stays in its AGPRs and the
    // compiler's own LDS wait in front of the MFMAs only covers the OLDER fragment set
    // (sched_barrier: the compiler otherwise sinks this step's MFMAs below the NEXT step's barrier and LDS wait,
    //  which puts the fragment-read latency back in front of them)
    auto step_steady = [&](int kt, const Frag& cur, Frag& nxt) {
      vm_wait_n<8>();
      lgkm_wait0();
      bare_barrier();
      issue(kt + 4);
      read_frags(kt + 1, nxt);
      __builtin_amdgcn_sched_barrier(0);
      mfma16(cur);
      __builtin_amdgcn_sched_barrier(0);
    };
    Frag f0, f1;
    read_frags(0, f0);
    int kt = 0;
    for (; kt + 5 < nk; kt += 2) {
      step_steady(kt, f0, f1);
      step_steady(kt + 1, f1, f0);
    }
    for (; kt < nk; kt += 2) {                 // drain: at most 6 slices
      step(kt, f0, f1);
      if (kt + 1 < nk) step(kt + 1, f1, f0);
    }
    if (s0 == 0 && s1 == S) {                  // whole tile: finished result
      const int64_t col = j0 + wn * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < I && col < J) {
          float v = acc[r];
          if (bias) v += bias[col];
          if (keep) v = keep[row * J + col] ? v * scale : 0.f;
          C[row * ldc + col] = v;
        }
      }
    } else {                                                // partial range: raw accumulator image, thread-major float4s
      const int seg = (u == u_begin) ? 0 : 1;
      float4* P = reinterpret_cast<float4*>(partials + ((size_t)blockIdx.x * 2 + seg) * kSkTileFloats);
#pragma unroll
      for (int q = 0; q < 4; ++q) P[q * kBlock + tid] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
    u += nk;
  }
}

// one block per output tile: adds the tile's partial slots in block order, applies bias + dropout, stores.
__global__ __launch_bounds__(kBlock) void sk_reduce_kernel(const float* __restrict__ partials, int tiles_j, int S,
                                                           int64_t total_units, int upb, int64_t I, int64_t J,
                                                           float* __restrict__ C, int64_t ldc,
                                                           const float* __restrict__ bias,
                                                           const uint8_t* __restrict__ keep, float scale) {
  const int64_t tile = blockIdx.x;
  const int64_t u_lo = tile * S, u_hi = u_lo + S;
  const int64_t b_first = u_lo / upb, b_last = (u_hi - 1) / upb;
  if (b_first == b_last) return;               // one block covered the whole tile and stored it itself
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float4 v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t b = b_first; b <= b_last; ++b) {
    const int seg = (b * upb < u_lo) ? 1 : 0;
    const float4* P = reinterpret_cast<const float4*>(partials + ((size_t)b * 2 + seg) * kSkTileFloats);
    float4 p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = P[q * kBlock + tid];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v[q].x += p[q].x; v[q].y += p[q].y; v[q].z += p[q].z; v[q].w += p[q].w;
    }
  }
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t i0 = (tile / tiles_j) * BT, j0 = (tile % tiles_j) * BT;
  const int64_t col = j0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r = 4 * q + c;
      const int64_t row = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (row < I && col < J) {
        float x = e[c];
        if (bias) x += bias[col];
        if (keep) x = keep[row * J + col] ? x * scale : 0.f;
        C[row * ldc + col] = x;
      }
    }
  }
}

// G [M, N] (optionally dropout-masked) -> T [N, Mp] fp32 transposed, zero in columns M..Mp-1, plus per-block column
// sums of the masked G (-> bias gradient): the A operand of the weight-gradient product gW = gY^T . F when that
// product runs through the forward kernel against a transposed copy of the constant feature matrix.
__global__ __launch_bounds__(kBlock) void transpose_mask_kernel(const float* __restrict__ G,
                                                                const uint8_t* __restrict__ keep, float scale,
                                                                int64_t M, int N, int64_t Mp, float* __restrict__ T,
                                                                float* __restrict__ colpart) {
  __shared__ float tile[64][65];
  __shared__ float red[4][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * 64;
  const int n0 = (int)blockIdx.y * 64;
  float sum = 0.f;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int r = q + 4 * i;
    const int64_t m = m0 + r;
    float v = 0.f;
    if (m < M && n0 + c < N) {
      v = G[m * N + n0 + c];
      if (keep) v = keep[m * N + n0 + c] ? v * scale : 0.f;
    }
    tile[r][c] = v;
    sum += v;
  }
  red[q][c] = sum;
  __syncthreads();
  if (q == 0 && n0 + c < N) colpart[(int64_t)blockIdx.x * N + n0 + c] = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int n = q + 4 * i;
    if (n0 + n < N && m0 + c < Mp) T[(int64_t)(n0 + n) * Mp + m0 + c] = tile[c][n];
  }
}

// ======================================================================================
// v10 weight gradient (default): gW[n][k] = sum_m G[m][n] F[m][k] with G = dropout-backward(gY), register-direct.
//
// The reduction runs over ROWS of the row-major operands, so the MFMA fragments can be loaded from global memory
// with fully coalesced 16-byte loads and no LDS: for v_mfma_f32_16x16x4_f32 lane l = (g = l / 16, j = l % 16) holds
// A[i = j][k = g] and B[k = g][col = j]. Lane (g, j) loads F[m + g][c0 + 4j .. 4j+3] and gY[m + g][n0 + 4j .. 4j+3]:
// one instruction covers 4 rows x 256 contiguous bytes, and component c of the two float4 is the operand of the
// MFMAs of column block c / channel block c, where block c holds the columns (channels) {4j + c}: the permutation
// of the OUTPUT index is undone for free when the tile is stored. Per step (4 rows per wave, 16 per block) a wave
// issues 3 loads (F, gY, 4 mask bytes) and 16 MFMAs on 16 different accumulators; 8 steps of loads are in flight
// in a register ring. Dropout backward and the bias gradient (column sums of G) happen on the loaded fragment:
// no masked copy of gY, no column-sum kernels. Block = 64 channels x 64 columns x one row range (`sp` ranges);
// the four waves' accumulators meet in LDS in wave order, row ranges in wg10_reduce_kernel in range order.
// Preconditions (host-checked): K % 64 == 0, N % 64 == 0.
// ======================================================================================
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <bool KEEP, bool NT, int WG_D>
__global__ __launch_bounds__(kBlock, WG_D <= 8 ? 2 : 1) void wgrad10_kernel(const float* __restrict__ gY,
                                                            const uint8_t* __restrict__ keep, float scale,
                                                            const float* __restrict__ F, int64_t M, int K, int N,
                                                            int64_t ms, float* __restrict__ out, int64_t split_stride,
                                                            float* __restrict__ bpart) {
  __shared__ __attribute__((aligned(16))) float4 red[4 * 1024];        // 4 waves x 16 blocks x 64 lanes: 64 KB
  __shared__ float bred[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int c0 = (int)blockIdx.x * 64, n0 = (int)blockIdx.y * 64, split = (int)blockIdx.z;
  const int64_t m_lo = (int64_t)split * ms, m_hi = min(M, m_lo + ms);
  const int nsteps = m_hi > m_lo ? (int)((m_hi - m_lo + 15) / 16) : 0;
  const float sc = KEEP ? scale : 1.f;
  floatx4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
  float bs[4] = {0.f, 0.f, 0.f, 0.f};
  // Register ring of WG_D steps, loads and waits left to the compiler. History: the loads used to be inline asm with
  // hand-counted `s_waitcnt vmcnt` (plain loads were sunk next to their use). To the compiler the destination of such a
  // load is a value that exists once the asm has executed, so the allocator could COPY it before the hand-written wait:
  // a `v_mov_b64` of slot 0's gradient fragment at the loop back-edge read a register whose load was still in flight,
  // and the epilogue's lane index was written into a register a dead tail prefetch could still overwrite. Alone the
  // loads had landed by then (7 steps of MFMAs); beside the GCN chain's SpMMs, at configs[4]'s size, one run in three or
  // four was 6e-3 off (round 5; tools/vmcnt_check.py finds both in the old ISA and guards the remaining hand-counted
  // kernel). What kept plain loads from staying where they are written was not the scheduler alone: the loads of a
  // `const __restrict__` KERNEL ARGUMENT count as constant memory, are chained to the DAG's entry node and float freely
  // inside the block. Through a laundered base pointer they are ordinary loads, ordered against side effects, and
  // sched_barrier(0) is one: each step's loads stay between its two barriers, SIInsertWaitcnts counts vmcnt itself.
  typedef const __attribute__((address_space(1))) char* gptr;        // still known to be global memory (not flat)
  gptr Fq = (gptr)(uintptr_t)F, Gq = (gptr)(uintptr_t)gY, Kq = (gptr)(uintptr_t)keep;
  asm volatile("" : "+s"(Fq), "+s"(Gq), "+s"(Kq));
  floatx4 fv[WG_D], gv[WG_D];
  uint32_t kv[WG_D];
  const int64_t m_mine = m_lo + 4 * wave + g;
  const int64_t m_base = min(m_mine, M - 1);
  const int t_last = m_mine < M ? (int)((M - 1 - m_mine) / 16) : 0;             // later steps re-load this row
  const int t_valid = m_mine < m_hi ? (int)((m_hi - 1 - m_mine) / 16) : -1;      // last step that contributes
  // 32-bit byte offsets from uniform bases (host-checked: M * max(K, N) < 2^30), 24-bit multiplies
  const uint32_t oF = (uint32_t)((m_base * K + c0 + 4 * j) * 4), oG = (uint32_t)((m_base * N + n0 + 4 * j) * 4);
  const uint32_t oK = (uint32_t)(m_base * N + n0 + 4 * j);
  const uint32_t sF = 64u * (uint32_t)K, sG = 64u * (uint32_t)N, sK = 16u * (uint32_t)N;
  auto load = [&](int slot, int t) {
    const uint32_t tc = (uint32_t)min(t, t_last);
    const uint32_t a = oF + __umul24(tc, sF), b = oG + __umul24(tc, sG);
    typedef const __attribute__((address_space(1))) floatx4* gvec;
    typedef const __attribute__((address_space(1))) uint32_t* gword;
    if (NT) fv[slot] = __builtin_nontemporal_load((gvec)(Fq + a));
    else fv[slot] = *(gvec)(Fq + a);
    gv[slot] = *(gvec)(Gq + b);
    if (KEEP) kv[slot] = *(gword)(Kq + (oK + __umul24(tc, sK)));
  };
  auto consume = [&](int slot, int t) {
    // the slot's readers start HERE (a side-effecting statement, ordered with the barriers): without it the scaling
    // multiplies of all eight slots are hoisted to the top of the iteration and drain the ring there
    if (KEEP) asm volatile("" : "+v"(fv[slot]), "+v"(gv[slot]), "+v"(kv[slot]));
    else asm volatile("" : "+v"(fv[slot]), "+v"(gv[slot]));
    const float s1 = t <= t_valid ? sc : 0.f;
    float a[4];
    if (KEEP) {
      a[0] = (kv[slot] & 0x000000ffu) ? gv[slot][0] * s1 : 0.f;
      a[1] = (kv[slot] & 0x0000ff00u) ? gv[slot][1] * s1 : 0.f;
      a[2] = (kv[slot] & 0x00ff0000u) ? gv[slot][2] * s1 : 0.f;
      a[3] = (kv[slot] & 0xff000000u) ? gv[slot][3] * s1 : 0.f;
    } else {
      a[0] = gv[slot][0] * s1;
      a[1] = gv[slot][1] * s1;
      a[2] = gv[slot][2] * s1;
      a[3] = gv[slot][3] * s1;
    }
    const float f[4] = {fv[slot][0], fv[slot][1], fv[slot][2], fv[slot][3]};
#pragma unroll
    for (int c = 0; c < 4; ++c) bs[c] += a[c];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb], f[cb], acc[rb][cb], 0, 0, 0);
    // pin: the MFMAs of this step are issued HERE (they have no side effect: left alone they all sink below the loads)
#pragma unroll
    for (int rb = 0; rb < 4; rb += 2)
      asm volatile("" : "+v"(acc[rb][0]), "+v"(acc[rb][1]), "+v"(acc[rb][2]), "+v"(acc[rb][3]), "+v"(acc[rb + 1][0]),
                        "+v"(acc[rb + 1][1]), "+v"(acc[rb + 1][2]), "+v"(acc[rb + 1][3]));
  };
#pragma unroll
  for (int d = 0; d < WG_D; ++d) {
    load(d, d);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int t0 = 0; t0 < nsteps; t0 += WG_D) {
#pragma unroll
    for (int d = 0; d < WG_D; ++d) {
      consume(d, t0 + d);
      __builtin_amdgcn_sched_barrier(0);
      load(d, t0 + d + WG_D);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float4* mine = red + wave * 1024;
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
      mine[(rb * 4 + cb) * 64 + lane] = make_float4(acc[rb][cb][0], acc[rb][cb][1], acc[rb][cb][2], acc[rb][cb][3]);
  if (bpart != nullptr && blockIdx.x == 0) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bs[c] += __shfl_xor(bs[c], 16);
      bs[c] += __shfl_xor(bs[c], 32);
    }
    if (g == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) bred[wave][4 * j + c] = bs[c];
    }
  }
  __syncthreads();
  // wave w finishes channel block w: D[i = 4g + r][jj = j] of block (rb, cb) is gW[n0 + 4 i + rb][c0 + 4 jj + cb]
  float4 v[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int idx = (wave * 4 + cb) * 64 + lane;
    const float4 r0 = red[idx], r1 = red[1024 + idx], r2 = red[2048 + idx], r3 = red[3072 + idx];
    v[cb].x = ((r0.x + r1.x) + r2.x) + r3.x;
    v[cb].y = ((r0.y + r1.y) + r2.y) + r3.y;
    v[cb].z = ((r0.z + r1.z) + r2.z) + r3.z;
    v[cb].w = ((r0.w + r1.w) + r2.w) + r3.w;
  }
  float* o = out + (size_t)split * split_stride;
  {
    const int64_t nb = n0 + 16 * g + wave;
    float* q = o + c0 + 4 * j;
    *reinterpret_cast<float4*>(q + (nb + 0) * K) = make_float4(v[0].x, v[1].x, v[2].x, v[3].x);
    *reinterpret_cast<float4*>(q + (nb + 4) * K) = make_float4(v[0].y, v[1].y, v[2].y, v[3].y);
    *reinterpret_cast<float4*>(q + (nb + 8) * K) = make_float4(v[0].z, v[1].z, v[2].z, v[3].z);
    *reinterpret_cast<float4*>(q + (nb + 12) * K) = make_float4(v[0].w, v[1].w, v[2].w, v[3].w);
  }
  if (bpart != nullptr && blockIdx.x == 0 && tid < 64)
    bpart[(size_t)split * N + n0 + tid] = ((bred[0][tid] + bred[1][tid]) + bred[2][tid]) + bred[3][tid];
}

// adds the row ranges' partial gradients in range order (4 outputs per thread); the last block adds the bias partials
__global__ __launch_bounds__(kBlock) void wg10_reduce_kernel(const float* __restrict__ P, int sp, int64_t total,
                                                             float* __restrict__ gW, const float* __restrict__ bpart,
                                                             int N, float* __restrict__ gb) {
  const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (4 * q < total) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < sp; ++s) {
      const float4 p = *reinterpret_cast<const float4*>(P + (size_t)s * total + 4 * q);
      v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    *reinterpret_cast<float4*>(gW + 4 * q) = v;
  }
  if (gb != nullptr && blockIdx.x == gridDim.x - 1) {
    for (int n = threadIdx.x; n < N; n += kBlock) {
      float b = 0.f;
      for (int s = 0; s < sp; ++s) b += bpart[(size_t)s * N + n];
      gb[n] = b;
    }
  }
}

// out[e] = epilogue(sum_s P[s][e]);  e = row*J + col.  Split loads are issued four at a time
// (independent) so the loop is bandwidth- not latency-bound; the add order is fixed.
__global__ __launch_bounds__(kBlock) void splitk_reduce_kernel(const float* __restrict__ P, int splits,
                                                               int64_t total, int64_t J,
                                                               const float* __restrict__ bias,
                                                               const uint8_t* __restrict__ keep, float scale,
                                                               float* __restrict__ out) {
  const int64_t n4 = total >> 2;   // J % 4 == 0 -> total % 4 == 0
  const int64_t t4 = total >> 2;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4* p = reinterpret_cast<const float4*>(P) + i;
    float4 v = p[0];
    int s = 1;
    for (; s + 3 < splits; s += 4) {
      const float4 a = p[(int64_t)s * t4], b = p[(int64_t)(s + 1) * t4];
      const float4 c = p[(int64_t)(s + 2) * t4], d = p[(int64_t)(s + 3) * t4];
      v.x = (((v.x + a.x) + b.x) + c.x) + d.x;
      v.y = (((v.y + a.y) + b.y) + c.y) + d.y;
      v.z = (((v.z + a.z) + b.z) + c.z) + d.z;
      v.w = (((v.w + a.w) + b.w) + c.w) + d.w;
    }
    for (; s < splits; ++s) {
      const float4 a = p[(int64_t)s * t4];
      v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    const int64_t e = i << 2;
    if (bias) {
      const int64_t c = e % J;
      v.x += bias[c]; v.y += bias[c + 1]; v.z += bias[c + 2]; v.w += bias[c + 3];
    }
    if (keep) {
      const uchar4 k = reinterpret_cast<const uchar4*>(keep)[i];
      v.x = k.x ? v.x * scale : 0.f;
      v.y = k.y ? v.y * scale : 0.f;
      v.z = k.z ? v.z * scale : 0.f;
      v.w = k.w ? v.w * scale : 0.f;
    }
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

// column sums of G[M,N] (N <= 256, N % 4 == 0): thread (c, rr) = (tid % N, tid / N) adds rows
// rr, rr + R, ... of its block's range; the R row-groups are combined through LDS.
//   stage 1 -> part[blocks][N];  stage 2 (one block, same scheme over the partials) -> out[N]
constexpr int kColsumBlocks = 128;
__device__ __forceinline__ float masked(const float* __restrict__ G, const uint8_t* __restrict__ mk, float ms,
                                        int64_t o) {
  const float v = G[o];
  return mk ? (mk[o] ? v * ms : 0.f) : v;
}
__device__ __forceinline__ void colsum_body(const float* __restrict__ G, int64_t row0, int64_t row_step,
                                            int64_t M, int N, float* __restrict__ dst,
                                            const uint8_t* __restrict__ mk = nullptr, float ms = 1.f) {
  __shared__ float red[kBlock];
  const int R = kBlock / N;
  const int c = threadIdx.x % N, rr = threadIdx.x / N;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (rr < R) {
    int64_t m = row0 + rr;
    for (; m + 3 * row_step < M; m += 4 * row_step) {      // four independent loads in flight
      a0 += masked(G, mk, ms, m * N + c);
      a1 += masked(G, mk, ms, (m + row_step) * N + c);
      a2 += masked(G, mk, ms, (m + 2 * row_step) * N + c);
      a3 += masked(G, mk, ms, (m + 3 * row_step) * N + c);
    }
    for (; m < M; m += row_step) a0 += masked(G, mk, ms, m * N + c);
  }
  red[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (threadIdx.x < N) {
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += red[r * N + threadIdx.x];
    dst[threadIdx.x] = s;
  }
}
__global__ __launch_bounds__(kBlock) void colsum_stage1(const float* __restrict__ G, int64_t M, int N,
                                                        float* __restrict__ part,
                                                        const uint8_t* __restrict__ mk, float ms) {
  const int R = kBlock / N;
  colsum_body(G, (int64_t)blockIdx.x * R, (int64_t)gridDim.x * R, M, N, part + (int64_t)blockIdx.x * N, mk, ms);
}
__global__ __launch_bounds__(kBlock) void colsum_stage2(const float* __restrict__ part, int nparts, int N,
                                                        float* __restrict__ out) {
  colsum_body(part, 0, kBlock / N, nparts, N, out);
}


// stream-K decomposition (gemm_sk_kernel): two equal unit ranges per CU, at least min(S, 8) slices each
struct SkPlan {
  int64_t tiles_i, tiles_j, total;
  int S, upb, blocks;
};
constexpr int kSkSlots = 512;
inline SkPlan sk_plan(int64_t I, int64_t J, int64_t KK) {
  SkPlan p;
  p.tiles_i = (I + BT - 1) / BT;
  p.tiles_j = (J + BT - 1) / BT;
  p.S = (int)(KK / BK);
  p.total = p.tiles_i * p.tiles_j * p.S;
  int64_t upb = (p.total + kSkSlots - 1) / kSkSlots;
  const int64_t floor_ = p.S < 8 ? p.S : 8;
  if (upb < floor_) upb = floor_;
  p.upb = (int)upb;
  p.blocks = (int)((p.total + upb - 1) / upb);
  return p;
}
inline bool sk_usable(int64_t KK) { return KK % BK == 0 && KK >= BK; }

// split count of the register-staged kernel: aim for >= ~4 blocks per CU, every split at least 4 slices deep
inline int choose_splits(int64_t tiles, int64_t KK) {
  const int64_t slices = (KK + BK - 1) / BK;
  int64_t s = (1024 + tiles - 1) / tiles;
  const int64_t max_s = slices / 4 > 0 ? slices / 4 : 1;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return (int)s;
}
inline int64_t chunk_for(int64_t KK, int splits) {
  const int64_t slices = (KK + BK - 1) / BK;
  return ((slices + splits - 1) / splits) * BK;
}

}  // namespace

extern "C" size_t mmssl_linear_workspace_bytes(int64_t M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 16;
  if (sk_usable(K)) return (size_t)sk_plan(M, N, K).blocks * 2 * kSkTileFloats * sizeof(float) + 16;
  const int64_t tiles = ((M + BT - 1) / BT) * ((N + BT - 1) / BT);
  const int splits = choose_splits(tiles, K);
  return splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 16;
}

extern "C" int mmssl_linear_f32(const float* F, const float* W, const float* b, const uint8_t* keep, float scale,
                                int64_t M, int K, int N, float* Y, void* workspace, size_t workspace_bytes,
                                void* stream) {
  if (M < 0 || K <= 0 || N <= 0 || (M > 0 && (!F || !W || !Y))) return MMSSL_E_BADARG;
  if ((K & 3) || (N & 3)) return MMSSL_E_UNSUPP;
  if (N > 256 && (b || keep || !sk_usable(K))) return MMSSL_E_UNSUPP;   // wide outputs: plain product only
  if (M == 0) return 0;
  if (((uintptr_t)F | (uintptr_t)W | (uintptr_t)Y) & 15) return MMSSL_E_BADARG;
  hipStream_t s = as_stream(stream);
  if (sk_usable(K)) {
    const SkPlan p = sk_plan(M, N, K);
    if (!workspace || workspace_bytes < (size_t)p.blocks * 2 * kSkTileFloats * sizeof(float)) return MMSSL_E_WORKSPACE;
    float* part = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL(gemm_sk_kernel, dim3((unsigned)p.blocks), dim3(kBlock), 0, s, F, (int64_t)K, W, (int64_t)K, M,
                       (int64_t)N, (int)p.tiles_j, p.S, p.total, p.upb, Y, (int64_t)N, b, keep, scale, part);
    MMSSL_LAUNCH_CHECK();
    if (p.upb % p.S != 0) {                    // some range ends inside a tile: partial slots exist
      hipLaunchKernelGGL(sk_reduce_kernel, dim3((unsigned)(p.tiles_i * p.tiles_j)), dim3(kBlock), 0, s, part,
                         (int)p.tiles_j, p.S, p.total, p.upb, M, (int64_t)N, Y, (int64_t)N, b, keep, scale);
      MMSSL_LAUNCH_CHECK();
    }
    return 0;
  }
  const int64_t tm = (M + BT - 1) / BT, tn = (N + BT - 1) / BT;
  const int splits = choose_splits(tm * tn, K);
  const int64_t chunk = chunk_for(K, splits);
  float* out = Y;
  if (splits > 1) {
    const size_t need = (size_t)splits * (size_t)M * (size_t)N * sizeof(float);
    if (!workspace || workspace_bytes < need) return MMSSL_E_WORKSPACE;
    out = reinterpret_cast<float*>(workspace);
  }
  // single split: bias + dropout in the kernel's own epilogue; otherwise in the fixed-order split reduce
  const float* kb = splits > 1 ? (const float*)nullptr : b;
  const uint8_t* kk = splits > 1 ? (const uint8_t*)nullptr : keep;
  const int64_t sstride = splits > 1 ? (int64_t)M * N : (int64_t)0;
  const dim3 grid((unsigned)tm, (unsigned)tn, (unsigned)splits);
  hipLaunchKernelGGL((gemm64_kernel<false>), grid, dim3(kBlock), 0, s, F, (int64_t)K, W, (int64_t)K, M, (int64_t)N,
                     (int64_t)K, chunk, out, (int64_t)N, sstride, kb, kk, scale, (const uint8_t*)nullptr, 1.f);
  MMSSL_LAUNCH_CHECK();
  if (splits > 1) {
    const int64_t total = M * N;
    int64_t nb = (total / 4 + kBlock - 1) / kBlock;
    nb = nb > 4096 ? 4096 : (nb < 1 ? 1 : nb);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(kBlock), 0, s, out, splits, total, (int64_t)N, b,
                       keep, scale, Y);
    MMSSL_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" size_t mmssl_transpose_mask_workspace_bytes(int64_t Mp, int N) {
  if (Mp <= 0 || N <= 0) return 16;
  return (size_t)((Mp + 63) / 64) * (size_t)N * sizeof(float) + 16;
}

extern "C" int mmssl_transpose_mask_f32(const float* G, const uint8_t* keep, float scale, int64_t M, int N, int64_t Mp,
                                        float* T, float* colsum, void* workspace, size_t workspace_bytes,
                                        void* stream) {
  if (M <= 0 || N <= 0 || !G || !T || Mp < M || (Mp & 3)) return MMSSL_E_BADARG;
  if ((N & 3) || N > 256) return MMSSL_E_UNSUPP;
  if (!workspace || workspace_bytes < mmssl_transpose_mask_workspace_bytes(Mp, N)) return MMSSL_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  float* colpart = reinterpret_cast<float*>(workspace);
  const unsigned nbm = (unsigned)((Mp + 63) / 64);
  hipLaunchKernelGGL(transpose_mask_kernel, dim3(nbm, (unsigned)((N + 63) / 64)), dim3(kBlock), 0, s, G, keep, scale, M,
                     N, Mp, T, colpart);
  MMSSL_LAUNCH_CHECK();
  if (colsum) {
    hipLaunchKernelGGL(colsum_stage2, dim3(1), dim3(kBlock), 0, s, colpart, (int)nbm, N, colsum);
    MMSSL_LAUNCH_CHECK();
  }
  return 0;
}

namespace {
// v10 decomposition: 64x64 output tiles x `sp` row ranges of at least 128 rows, about two blocks per CU. Measured for the
// Baby image / text shapes: alone, 256 blocks 93 / 34 us and 512 blocks 94 / 46 us; inside the hot-path step (next to the
// SpMM chains) 512 blocks 0.610-0.620 ms per step, 256 blocks 0.625 ms - a second resident block covers the load latency
// that grows when the SpMMs share the CU - so the default follows the step.
struct WgPlan {
  int tk, tn, sp;
  int64_t ms;
};
inline bool wg10_usable(int64_t M, int K, int N) {
  return K % 64 == 0 && N % 64 == 0 && M * (int64_t)(K > N ? K : N) < ((int64_t)1 << 30) &&
         K < (1 << 18) && N < (1 << 18);
}
inline WgPlan wg10_plan(int64_t M, int K, int N) {
  // (max_sp was 8 until round 6: configs[4]'s [1M, 128] x [1M, 128] then ran as 2 x 2 x 8 = 32 blocks on 256 CUs)
  constexpr int target = 512, max_sp = 128;
  WgPlan p;
  p.tk = K / 64;
  p.tn = N / 64;
  const int64_t tiles = (int64_t)p.tk * p.tn;
  int64_t sp = ((target > 0 ? target : 512) + tiles - 1) / tiles;
  const int64_t cap = M / 128 > 0 ? M / 128 : 1;
  if (sp > cap) sp = cap;
  if (sp > max_sp) sp = max_sp;
  if (sp < 1) sp = 1;
  p.ms = ((M + sp - 1) / sp + 15) / 16 * 16;
  p.sp = (int)((M + p.ms - 1) / p.ms);
  return p;
}
}  // namespace

extern "C" size_t mmssl_linear_wgrad_workspace_bytes(int64_t M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 16;
  if (wg10_usable(M, K, N)) {
    const WgPlan p = wg10_plan(M, K, N);
    return ((size_t)p.sp * (size_t)N * (size_t)K + (size_t)p.sp * (size_t)N) * sizeof(float) + 16;
  }
  const int splits = choose_splits(((N + BT - 1) / BT) * ((K + BT - 1) / BT), M);
  const size_t part = splits > 1 ? (size_t)splits * (size_t)N * (size_t)K * sizeof(float) : 0;
  return part + (size_t)kColsumBlocks * (size_t)N * sizeof(float) + 16;
}

extern "C" int mmssl_linear_wgrad_fuses_mask(int64_t M, int K, int N) {
  return (M > 0 && K > 0 && N > 0 && wg10_usable(M, K, N)) ? 1 : 0;
}
extern "C" int mmssl_linear_wgrad_f32(const float* gY, const uint8_t* keep, float scale, const float* F, int64_t M,
                                      int K, int N, float* gW, float* gb, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  if (M <= 0 || K <= 0 || N <= 0 || !gY || !F || !gW) return MMSSL_E_BADARG;
  if ((K & 3) || (N & 3) || N > 256) return MMSSL_E_UNSUPP;
  if (!workspace || workspace_bytes < mmssl_linear_wgrad_workspace_bytes(M, K, N)) return MMSSL_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  if (wg10_usable(M, K, N)) {
    if (((uintptr_t)gY | (uintptr_t)F | (uintptr_t)gW | (uintptr_t)keep) & 3) return MMSSL_E_BADARG;
    if (((uintptr_t)gY | (uintptr_t)F | (uintptr_t)gW) & 15) return MMSSL_E_BADARG;
    const WgPlan p = wg10_plan(M, K, N);
    float* P = reinterpret_cast<float*>(workspace);      // [sp][N][K]
    float* bpart = P + (size_t)p.sp * N * K;             // [sp][N]
    const bool direct = p.sp == 1;
    // 8 steps of loads in flight (a ring of 12 measured 172 vs 93 us: the accumulators no longer fit next to it)
    auto* kern = keep ? wgrad10_kernel<true, false, 8> : wgrad10_kernel<false, false, 8>;
    hipLaunchKernelGGL(kern, dim3((unsigned)p.tk, (unsigned)p.tn, (unsigned)p.sp), dim3(kBlock), 0, s, gY, keep, scale, F,
                       M, K, N, p.ms, direct ? gW : P, (int64_t)N * K, gb ? (direct ? gb : bpart) : (float*)nullptr);
    MMSSL_LAUNCH_CHECK();
    if (!direct) {
      const int64_t total = (int64_t)N * K;
      const int64_t nb = (total / 4 + kBlock - 1) / kBlock;
      hipLaunchKernelGGL(wg10_reduce_kernel, dim3((unsigned)nb), dim3(kBlock), 0, s, P, p.sp, total, gW, bpart, N, gb);
      MMSSL_LAUNCH_CHECK();
    }
    return 0;
  }
  const int64_t tn = (N + BT - 1) / BT, tk = (K + BT - 1) / BT;
  const int splits = choose_splits(tn * tk, M);
  const int64_t chunk = chunk_for(M, splits);
  float* ws = reinterpret_cast<float*>(workspace);
  float* colpart = ws;                                   // [kColsumBlocks][N]
  float* P = ws + (size_t)kColsumBlocks * N;             // [splits][N][K]
  // gW[n][k] = sum_m gY[m][n] * F[m][k]: A = gY as [kk=m][i=n], B = F as [kk=m][j=k]
  hipLaunchKernelGGL((gemm64_kernel<true>), dim3((unsigned)tn, (unsigned)tk, (unsigned)splits), dim3(kBlock), 0, s, gY,
                     (int64_t)N, F, (int64_t)K, (int64_t)N, (int64_t)K, M, chunk, splits == 1 ? gW : P, (int64_t)K,
                     splits == 1 ? (int64_t)0 : (int64_t)N * K, (const float*)nullptr, (const uint8_t*)nullptr, 1.f,
                     keep, scale);
  MMSSL_LAUNCH_CHECK();
  if (splits > 1) {
    const int64_t total = (int64_t)N * K;
    int64_t nb = (total / 4 + kBlock - 1) / kBlock;
    nb = nb > 4096 ? 4096 : (nb < 1 ? 1 : nb);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(kBlock), 0, s, P, splits, total, (int64_t)K,
                       (const float*)nullptr, (const uint8_t*)nullptr, 1.f, gW);
    MMSSL_LAUNCH_CHECK();
  }
  if (gb) {
    const int R = kBlock / N;
    int64_t nb = (M + R - 1) / R;
    nb = nb > kColsumBlocks ? kColsumBlocks : nb;
    hipLaunchKernelGGL(colsum_stage1, dim3((unsigned)nb), dim3(kBlock), 0, s, gY, M, N, colpart, keep, scale);
    MMSSL_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_stage2, dim3(1), dim3(kBlock), 0, s, colpart, (int)nb, N, gb);
    MMSSL_LAUNCH_CHECK();
  }
  return 0;
}
