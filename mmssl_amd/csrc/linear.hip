// Modality projection GEMM on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32:
// exact fp32, 157 TF peak = the only MFMA use on this path; everything else is HBM-bound).
// Replaces nn.Linear image_trans / text_trans + nn.Dropout and their autograd
// (/root/reference/MMSSL/Models.py:28-29, 54, 173-174).
//
//   forward : Y[M,N]  = dropout(F[M,K] . W[N,K]^T + b)          (F streamed once from HBM)
//   wgrad   : gW[N,K] = gY[M,N]^T . F[M,K],  gb[N] = colsum(gY)
//
// Tiling (both kernels): a 256-thread block (2x2 waves, one 32x32 MFMA accumulator each) owns a
// 64x64 output tile and walks its reduction range in 32-deep slices; the reduction dimension is
// split over blockIdx.z so that >= ~4 blocks per CU exist even for M = 18K; split partials are
// summed in a fixed order by a small epilogue kernel that also applies bias + dropout
// (deterministic, no float atomics).
//   gemm64_kernel<DIRECT>  register-staged: slices fetched global->registers one slice ahead,
//       written to a double-buffered k-major LDS image (row stride 65: conflict-free transposed
//       ds_write_b32 and fragment ds_read_b32), one barrier per slice. Used for wgrad (both
//       operand layouts, ragged reduction length, dropout mask applied while fetching gY) and as the
//       forward fallback for K ranges that are not whole slices.
//   gemm_fwd_dma_kernel    forward default: slices go global->LDS by LDS-DMA into a 4-stage ring
//       (XOR-swizzled through the source addresses), fragments come back as ds_read_b128.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"

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


// ======================================================================================
// v5 forward kernel: same 64x64 block tile / 32x32-per-wave MFMA tiling, but the operand slices go
// global -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass) into a
// 4-stage ring (3 slices = 24 KB of F in flight per block), and the fragments come back with
// ds_read_b128. Decomposition runs of v4 (MMSSL_GEMM_MODE) showed the forward is bound by what returns
// INTO the VGPRs: VMEM load returns and LDS fragment reads add up (stream-only 55 us, MFMA+LDS-only
// 70 us, both 110 us, unchanged without LDS stores or without barriers). LDS-DMA removes the first term.
//
// LDS image of one operand slice: 64 rows x 8 chunks of 16 B, chunk c of row i at slot
// i*8 + (c ^ ((i >> 1) & 7)). LDS-DMA writes lane-linearly (M0 base + lane*16), so the XOR swizzle is
// applied to each lane's SOURCE address; the same involution is applied when reading. With it the 16-lane
// groups of ds_read_b128 ({0-3,12-15,20-27}, ...) cover all 64 banks exactly once.
// A lane's float4 (4 consecutive k) feeds 4 MFMA steps; step 4q+e contracts k = {8q+e, 8q+4+e}
// (the pairing of k indices is free as long as A and B agree).
// Preconditions (host-checked): chunk % 32 == 0, KK % chunk == 0, chunk >= 96, row-major [i][kk] operands.
// ======================================================================================
constexpr int kDmaStages = 4;
constexpr int kDmaStageFloats = 2 * BT * BK;        // A slice then B slice: 8 KB + 8 KB

// same with the non-temporal hint: the streamed operand (the feature matrix, read once per product) should not push the
// SpMM chains' tables out of L2 while both run side by side
__device__ __forceinline__ void glds16_nt(const float* gsrc, unsigned lds_dst) {
  unsigned keep_m0;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
      : "=&s"(keep_m0)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep_m0;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep_m0)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait_n() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void bare_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(kBlock) void gemm_fwd_dma_kernel(const float* __restrict__ A, int64_t lda,
                                                              const float* __restrict__ B, int64_t ldb, int64_t I,
                                                              int64_t J, int64_t kk_chunk, float* __restrict__ C,
                                                              int64_t ldc, int64_t split_stride,
                                                              const float* __restrict__ bias,
                                                              const uint8_t* __restrict__ keep, float scale) {
  __shared__ __attribute__((aligned(16))) float ring[kDmaStages * kDmaStageFloats];     // 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t i0 = (int64_t)blockIdx.x * BT, j0 = (int64_t)blockIdx.y * BT;
  const int64_t kk_beg = (int64_t)blockIdx.z * kk_chunk;
  const int nk = (int)(kk_chunk / BK);
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // DMA pieces: wave w moves rows [16w, 16w+16) of each operand slice as 2 x (8 rows x 128 B)
  const float* pa[2];
  const float* pb[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = 16 * wave + 8 * j + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);                  // source-side swizzle
    pa[j] = A + min(i0 + r, I - 1) * lda + kk_beg + 4 * c;
    pb[j] = B + min(j0 + r, J - 1) * ldb + kk_beg + 4 * c;
  }
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  const unsigned piece = __builtin_amdgcn_readfirstlane((unsigned)(16 * wave * BK * 4));
  auto issue = [&](int kt) {
    const unsigned st = ring_lds + (unsigned)(kt & (kDmaStages - 1)) * (kDmaStageFloats * 4) + piece;
    glds16(pa[0] + (int64_t)kt * BK, st);
    glds16(pa[1] + (int64_t)kt * BK, st + 8 * BK * 4);
    glds16(pb[0] + (int64_t)kt * BK, st + BT * BK * 4);
    glds16(pb[1] + (int64_t)kt * BK, st + BT * BK * 4 + 8 * BK * 4);
  };
  issue(0);
  issue(1);
  issue(2);
  const int h = lane >> 5, lr = lane & 31;
  const int sw = (lr >> 1) & 7;
  const int ia = (wm * 32 + lr) * BK, jb = BT * BK + (wn * 32 + lr) * BK;
  for (int kt = 0; kt < nk; ++kt) {
    // this wave's pieces of slice kt have landed when at most the pieces of the slices issued after it remain
    if (kt + 2 < nk) vm_wait_n<8>();
    else if (kt + 1 < nk) vm_wait_n<4>();
    else vm_wait_n<0>();
    bare_barrier();          // everyone's pieces of slice kt are in LDS; stage (kt-1)&3 is no longer read
    if (kt + 3 < nk) issue(kt + 3);
    const float* st = ring + (kt & (kDmaStages - 1)) * kDmaStageFloats;
    float4 fa[4], fb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int pos = ((2 * q + h) ^ sw) * 4;
      fa[q] = *reinterpret_cast<const float4*>(st + ia + pos);
      fb[q] = *reinterpret_cast<const float4*>(st + jb + pos);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].x, fb[q].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].y, fb[q].y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].z, fb[q].z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].w, fb[q].w, acc, 0, 0, 0);
    }
  }
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


// ======================================================================================
// v6 "stream-K" forward kernel (default since round 2). Same 64x64 block tile, 2x2 waves, 32-deep slices and
// swizzled 4-stage LDS-DMA ring as gemm_fwd_dma_kernel, with two changes that the round-1 profile asked for:
//
//  * WORK DECOMPOSITION. The split-K grid of v5 launches tiles x splits blocks of equal size; for the Baby image
//    projection that is 287 x 4 = 1148 blocks on 512 block slots (2 per CU): 2.24 rounds, i.e. the last round
//    runs on a quarter of the chip. Here the (tile, slice) units are laid out on one axis and cut into as many
//    EQUAL contiguous ranges as there are block slots (287 x 128 = 36736 units / 512 = 71.75): every block does
//    the same number of slices, whatever the tile count. A range that covers a whole tile stores the result
//    directly (bias + dropout in the epilogue); partial ranges store the raw accumulator to one of the block's two
//    slots (head / tail) and sk_reduce_kernel adds a tile's slots in block order (fixed order: deterministic).
//    Partial traffic: <= 2 x 16 KB per block (13 MB for the Baby image shape, vs 19 MB of split partials in v5).
//  * FRAGMENT DOUBLE BUFFERING. v5 read the 8 ds_read_b128 fragments of a slice right after the barrier and only
//    then started the slice's 16 dependent MFMAs, so every slice paid barrier + LDS latency in front of its MFMA
//    chain (PMC: SQ_WAIT_INST_ANY 66 % of wave cycles, MFMA pipe 59 % busy). Now the fragments of slice kt+1 are
//    fetched into a second register set BEFORE the MFMAs of slice kt are issued.
// Preconditions (host-checked): KK % 32 == 0, row-major [i][kk] operands, 16-B aligned rows.
// ======================================================================================
constexpr int kSkTileFloats = BT * BT;      // one partial slot: the block's 64x64 accumulator image (16 KB)

// lgkmcnt(0) as the BUILTIN (simm16: vmcnt = 63, expcnt = 7, lgkmcnt = 0): the compiler's wait-count pass sees it,
// so it does not put redundant s_waitcnt instructions between the dependent MFMAs that follow (any instruction
// between two MFMAs on one accumulator breaks their back-to-back issue).
__device__ __forceinline__ void lgkm_wait0() {
  __builtin_amdgcn_s_waitcnt(0xC07F);
  asm volatile("" ::: "memory");
}

struct Frag {
  float4 a[4], b[4];
};

__global__ __launch_bounds__(kBlock) void gemm_sk_kernel(const float* __restrict__ A, int64_t lda,
                                                         const float* __restrict__ B, int64_t ldb, int64_t I,
                                                         int64_t J, int tiles_j, int S, int64_t total_units, int upb,
                                                         float* __restrict__ C, int64_t ldc,
                                                         const float* __restrict__ bias,
                                                         const uint8_t* __restrict__ keep, float scale,
                                                         float* __restrict__ partials, int prio,
                                                         int* __restrict__ tickets) {
  __shared__ __attribute__((aligned(16))) float ring[kDmaStages * kDmaStageFloats];     // 64 KB
  __shared__ int s_last;
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
      if (prio & 2) {                       // MMSSL_GEMM_NT=1: stream the A operand past L2
        glds16_nt(pa[0] + (int64_t)kt * BK, st);
        glds16_nt(pa[1] + (int64_t)kt * BK, st + 8 * BK * 4);
      } else {
        glds16(pa[0] + (int64_t)kt * BK, st);
        glds16(pa[1] + (int64_t)kt * BK, st + 8 * BK * 4);
      }
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
      if (prio & 1) __builtin_amdgcn_s_setprio(3);     // MMSSL_GEMM_PRIO=1 (experiment): MFMA chain outranks the partner wave
      mfma16(cur);
      if (prio & 1) __builtin_amdgcn_s_setprio(0);
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
    } else if (tickets == nullptr) {            // partial range: raw accumulator image, thread-major float4s
      const int seg = (u == u_begin) ? 0 : 1;
      float4* P = reinterpret_cast<float4*>(partials + ((size_t)blockIdx.x * 2 + seg) * kSkTileFloats);
#pragma unroll
      for (int q = 0; q < 4; ++q) P[q * kBlock + tid] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    } else {
      // In-kernel fix-up: the LAST block to deliver a tile's partial adds all of them in block order (the order
      // sk_reduce_kernel uses: the result does not depend on who is last) and runs the epilogue; no second kernel
      // stands between the product and its consumer. The images travel as write-through (sc1) stores and sc1 loads,
      // like the SpMM's in-kernel combine (graph.hip): no release/acquire fence, other dirty lines stay in this L2.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "gemm_sk_kernel's in-kernel fix-up is written for gfx942/gfx950 memory semantics (stores retire through vmcnt)"
#endif
      typedef unsigned long long u64;
      const int seg = (u == u_begin) ? 0 : 1;
      u64* P = reinterpret_cast<u64*>(partials + ((size_t)blockIdx.x * 2 + seg) * kSkTileFloats);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        u64 v;
        const float2 f2 = make_float2(acc[2 * q], acc[2 * q + 1]);
        __builtin_memcpy(&v, &f2, 8);
        __hip_atomic_store(P + q * kBlock + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const int64_t u_lo = tile * S;
      const int64_t b_first = u_lo / upb, b_last = (u_lo + S - 1) / upb;
      if (tid == 0) {
        const int prev = __hip_atomic_fetch_add(tickets + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = prev == (int)(b_last - b_first);
        if (last) __hip_atomic_store(tickets + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // re-arm
        s_last = last;
      }
      __syncthreads();
      if (s_last) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = 0.f;
        for (int64_t b = b_first; b <= b_last; ++b) {
          const int sg = (b * upb < u_lo) ? 1 : 0;
          const u64* Q = reinterpret_cast<const u64*>(partials + ((size_t)b * 2 + sg) * kSkTileFloats);
          u64 w[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) w[q] = __hip_atomic_load(Q + q * kBlock + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            float2 f2;
            __builtin_memcpy(&f2, &w[q], 8);
            v[2 * q] += f2.x;
            v[2 * q + 1] += f2.y;
          }
        }
        const int64_t col = j0 + wn * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          if (row < I && col < J) {
            float x = v[r];
            if (bias) x += bias[col];
            if (keep) x = keep[row * J + col] ? x * scale : 0.f;
            C[row * ldc + col] = x;
          }
        }
      }
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
// v7 "ping-pong" kernel: the stream-K decomposition of v6 on a 128x64 tile with EIGHT waves per block that
// alternate roles. v5/v6 counters (profiles/r02_gemm_pmc_*.txt): MFMA pipe 59 % busy, waves 66 % of their time
// inside a dependent MFMA chain — with two unsynchronised waves per SIMD the chains of both collide on the one
// matrix pipe while at other times both are in their DMA / LDS / barrier part and the pipe idles. Here the two
// waves of a SIMD belong to ONE block and are anti-synchronised by its barrier: during phase p group p%2 runs the
// 32 MFMAs of slice p from registers while the other group issues the LDS-DMA of slice p+4, reads the fragments of
// slice p+1 into registers and waits for slice p+2 to land; then the roles swap. A wave covers 32 rows x 64
// columns with two independent accumulators (the A fragment is used twice: 12 instead of 16 ds_read_b128 per
// 32 MFMAs, and W is fetched once per 128 rows of F instead of once per 64). The two groups' accumulators (even /
// odd slices) are added through LDS at the end of a range. 6-stage ring of 24 KB (16 KB A + 8 KB B) = 144 KB,
// one block per CU, ranges of equal length (stream-K) over 256 blocks.
// ======================================================================================
constexpr int PT_I = 128;                    // tile rows
constexpr int kPpThreads = 512;
constexpr int kPpTileFloats = PT_I * BT;     // one partial slot (32 KB)

template <int N>
__device__ __forceinline__ void vm_wait_groups(int cnt) {     // s_waitcnt vmcnt(N * cnt), cnt in 0..4
  switch (cnt) {
    case 0: vm_wait_n<0>(); break;
    case 1: vm_wait_n<N>(); break;
    case 2: vm_wait_n<2 * N>(); break;
    case 3: vm_wait_n<3 * N>(); break;
    default: vm_wait_n<4 * N>(); break;
  }
}

// PBK = slice depth (16 or 32 floats), NS = ring stages, D = issue distance (even, D + 2 <= NS): slice p + D is
// issued in phase p and must have landed by the end of phase p + D - 2. The decomposition runs
// (profiles/r02_gemm_modes.txt) showed the first version (PBK 32, 6 x 24 KB, D = 4: two phases = 1.8 us of lead)
// LATENCY-bound - the DMA-only run needs 54 us at ~3 us of loaded HBM latency - so the default is PBK 16 with
// 12 x 12 KB stages and D = 10: the same LDS, eight phases (~3.8 us) of lead.
template <int PBK, int NS, int D>
__global__ __launch_bounds__(kPpThreads) void gemm_pp_kernel(const float* __restrict__ A, int64_t lda,
                                                             const float* __restrict__ B, int64_t ldb, int64_t I,
                                                             int64_t J, int tiles_j, int S, int64_t total_units,
                                                             int upb, float* __restrict__ C, int64_t ldc,
                                                             int transpose_out, const float* __restrict__ bias,
                                                             const uint8_t* __restrict__ keep, float scale,
                                                             float* __restrict__ partials, int dbg) {
  // dbg (tools/gemm_mode_probe.py only; 0 in production): bit 0 = no LDS-DMA, bit 1 = no MFMA, bit 2 = no fragment reads
  constexpr int CPR = PBK / 4;                       // 16-B chunks per row
  constexpr int RPP = 256 / PBK;                     // rows per 1-KB DMA piece
  constexpr int PA = PT_I / RPP / 4, PB = BT / RPP / 4;       // pieces per wave of the loading group
  constexpr int OPS = PA + PB;                       // DMA instructions per wave per slice
  constexpr int NQ = PBK / 8;                        // float4 fragments per operand per slice
  constexpr int STAGE = (PT_I + BT) * PBK;           // floats
  static_assert(D % 2 == 0 && D + 2 <= NS && (D - 2) / 2 <= 4, "ring geometry");
  extern __shared__ __attribute__((aligned(16))) float ring[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = wave >> 2, w = wave & 3;                 // role group, wave within the group
  const int h = lane >> 5, lr = lane & 31;
  auto swz = [](int r) { return PBK == 32 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
  const int sw = swz(lr);                                // rows w*32 + lr and lr (+32) share lr's swizzle
  const int ia = (w * 32 + lr) * PBK;
  const int jb0 = PT_I * PBK + lr * PBK, jb1 = jb0 + 32 * PBK;
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  const int64_t u_begin = (int64_t)blockIdx.x * upb;
  const int64_t u_end = min(total_units, u_begin + upb);
  int64_t u = u_begin;
  while (u < u_end) {
    const int64_t tile = u / S;
    const int s0 = (int)(u - tile * S);
    const int s1 = (int)min((int64_t)S, s0 + (u_end - u));
    const int nk = s1 - s0;
    const int64_t i0 = (tile / tiles_j) * PT_I, j0 = (tile % tiles_j) * BT;
    // DMA pieces of one stage, issued by the 4 waves of the loading group: A rows [32w, 32w+32), B rows [16w, 16w+16)
    const float* pa[PA];
    const float* pb[PB];
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      const int r = 32 * w + RPP * j + lane / CPR;
      const int c = (lane % CPR) ^ swz(r);
      pa[j] = A + min(i0 + r, I - 1) * lda + (int64_t)s0 * PBK + 4 * c;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int r = 16 * w + RPP * j + lane / CPR;
      const int c = (lane % CPR) ^ swz(r);
      pb[j] = B + min(j0 + r, J - 1) * ldb + (int64_t)s0 * PBK + 4 * c;
    }
    const unsigned dst_a = __builtin_amdgcn_readfirstlane((unsigned)(32 * w * PBK * 4));
    const unsigned dst_b = __builtin_amdgcn_readfirstlane((unsigned)((PT_I * PBK + 16 * w * PBK) * 4));
    auto issue = [&](int kt) {
      if (dbg & 1) return;
      const unsigned st = ring_lds + (unsigned)(kt % NS) * (STAGE * 4);
#pragma unroll
      for (int j = 0; j < PA; ++j) glds16(pa[j] + (int64_t)kt * PBK, st + dst_a + j * 1024);
#pragma unroll
      for (int j = 0; j < PB; ++j) glds16(pb[j] + (int64_t)kt * PBK, st + dst_b + j * 1024);
    };
    float4 fa[NQ], fb0[NQ], fb1[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) fa[q] = fb0[q] = fb1[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto read_frags = [&](int kt) {
      if (dbg & 4) return;
      const float* st = ring + (kt % NS) * STAGE;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int pos = ((2 * q + h) ^ sw) * 4;
        fa[q] = *reinterpret_cast<const float4*>(st + ia + pos);
        fb0[q] = *reinterpret_cast<const float4*>(st + jb0 + pos);
        fb1[q] = *reinterpret_cast<const float4*>(st + jb1 + pos);
      }
    };
    floatx16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    auto compute = [&]() {
      if (dbg & 2) return;
      if (dbg & 8) __builtin_amdgcn_s_setprio(3);        // experiment: the computing wave outranks its SIMD partner
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].x, fb0[q].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].x, fb1[q].x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].y, fb0[q].y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].y, fb1[q].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].z, fb0[q].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].z, fb1[q].z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].w, fb0[q].w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q].w, fb1[q].w, acc1, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (dbg & 8) __builtin_amdgcn_s_setprio(0);
    };
    // loading role of phase p (p may be -1): fragments of slice p+1, slice p+2 must have landed, slice p+D is issued
    auto load_steady = [&](int p) {            // requires p + D < nk
      issue(p + D);
      read_frags(p + 1);
      vm_wait_n<OPS*(D - 2) / 2>();
      lgkm_wait0();
    };
    auto load_any = [&](int p) {
      if (p + D < nk) issue(p + D);
      if (p + 1 < nk) read_frags(p + 1);
      if (p + 2 < nk) {
        int cnt = 0;                           // my groups issued after slice p+2: p+4, p+6, ..., p+D (those < nk)
#pragma unroll
        for (int t = 4; t <= D; t += 2) cnt += (p + t < nk) ? 1 : 0;
        vm_wait_groups<OPS>(cnt);
      }
      lgkm_wait0();
    };
    // ring hand-over from the previous range (its epilogue reads the ring, its stores use vmcnt)
    vm_wait_n<0>();
    lgkm_wait0();
    bare_barrier();
    {                                           // slices 0 .. D-2 in flight: even ones from group 1, odd from group 0
      int mine = 0;
#pragma unroll
      for (int t = 0; t <= D - 2; ++t)
        if ((t & 1) != g && t < nk) { issue(t); ++mine; }
      if (g == 1) vm_wait_groups<OPS>(mine > 0 ? mine - 1 : 0);      // slice 0 has landed (mine)
    }
    bare_barrier();
    if (g == 0) {
      load_any(-1);
      bare_barrier();
      int p = 0;
      for (; p + D + 2 < nk; p += 2) {
        compute();                              // slice p
        bare_barrier();
        load_steady(p + 1);
        bare_barrier();
      }
      for (; p < nk; p += 2) {
        compute();
        bare_barrier();
        if (p + 1 < nk) {
          load_any(p + 1);
          bare_barrier();
        }
      }
    } else {
      bare_barrier();                           // phase -1: nothing to do
      int p = 0;
      for (; p + D + 2 < nk; p += 2) {
        load_steady(p);
        bare_barrier();
        compute();                              // slice p + 1
        bare_barrier();
      }
      for (; p < nk; p += 2) {
        load_any(p);
        bare_barrier();
        if (p + 1 < nk) {
          compute();
          bare_barrier();
        }
      }
    }
    // add the two groups' accumulators (even + odd slices) through LDS; group 0 owns the result
    vm_wait_n<0>();
    float4* X = reinterpret_cast<float4*>(ring);
    const int t = tid & 255;
    if (g == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        X[q * 256 + t] = make_float4(acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]);
        X[(4 + q) * 256 + t] = make_float4(acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]);
      }
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 x0 = X[q * 256 + t], x1 = X[(4 + q) * 256 + t];
        acc0[4 * q] += x0.x; acc0[4 * q + 1] += x0.y; acc0[4 * q + 2] += x0.z; acc0[4 * q + 3] += x0.w;
        acc1[4 * q] += x1.x; acc1[4 * q + 1] += x1.y; acc1[4 * q + 2] += x1.z; acc1[4 * q + 3] += x1.w;
      }
      if (s0 == 0 && s1 == S) {                 // whole tile: finished result
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          const int64_t col = j0 + 32 * a + lr;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int64_t row = i0 + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < I && col < J) {
              float v = a ? acc1[r] : acc0[r];
              if (bias) v += bias[col];
              if (keep) v = keep[row * J + col] ? v * scale : 0.f;
              if (transpose_out) C[col * ldc + row] = v;
              else C[row * ldc + col] = v;
            }
          }
        }
      } else {
        const int seg = (u == u_begin) ? 0 : 1;
        float4* P = reinterpret_cast<float4*>(partials + ((size_t)blockIdx.x * 2 + seg) * kPpTileFloats);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          P[q * 256 + t] = make_float4(acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]);
          P[(4 + q) * 256 + t] = make_float4(acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]);
        }
      }
    }
    u += nk;
  }
}

// one block (256 threads) per 128x64 output tile: adds the tile's partial slots in block order
__global__ __launch_bounds__(kBlock) void pp_reduce_kernel(const float* __restrict__ partials, int tiles_j, int S,
                                                           int64_t total_units, int upb, int64_t I, int64_t J,
                                                           float* __restrict__ C, int64_t ldc, int transpose_out,
                                                           const float* __restrict__ bias,
                                                           const uint8_t* __restrict__ keep, float scale) {
  const int64_t tile = blockIdx.x;
  const int64_t u_lo = tile * S, u_hi = u_lo + S;
  const int64_t b_first = u_lo / upb, b_last = (u_hi - 1) / upb;
  if (b_first == b_last) return;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int h = lane >> 5, lr = lane & 31;
  float4 v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t b = b_first; b <= b_last; ++b) {
    const int seg = (b * upb < u_lo) ? 1 : 0;
    const float4* P = reinterpret_cast<const float4*>(partials + ((size_t)b * 2 + seg) * kPpTileFloats);
    float4 p[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) p[q] = P[q * 256 + t];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      v[q].x += p[q].x; v[q].y += p[q].y; v[q].z += p[q].z; v[q].w += p[q].w;
    }
  }
  const int64_t i0 = (tile / tiles_j) * PT_I, j0 = (tile % tiles_j) * BT;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int a = q >> 2;
    const int64_t col = j0 + 32 * a + lr;
    const float e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r = 4 * (q & 3) + c;
      const int64_t row = i0 + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < I && col < J) {
        float x = e[c];
        if (bias) x += bias[col];
        if (keep) x = keep[row * J + col] ? x * scale : 0.f;
        if (transpose_out) C[col * ldc + row] = x;
        else C[row * ldc + col] = x;
      }
    }
  }
}

// ======================================================================================
// v8 (MMSSL_GEMM_V=8, experiment): ONE wave per SIMD, 256x64 block tile, every wave a 64x64 tile with four
// independent accumulators, the slice's LDS-DMA issue and the next slice's fragment reads interleaved between the
// MFMAs of the current slice (one filler after every group of four MFMAs on four different accumulators).
// Why: the v7 decomposition shows each side near its own limit alone and the MFMA wave slowed by whatever its SIMD
// partner does; here there is no partner, and a 64x64 wave tile needs 1 fragment register per MFMA instead of 1.5 (v7)
// or 2 (v5/v6). 3 stages of 40 KB (A 256x32 + B 64x32 floats); a wave DMAs its OWN 64 A rows and a quarter of B.
// ======================================================================================
constexpr int W8_I = 256;
constexpr int kW8Stages = 3;
constexpr int kW8StageFloats = (W8_I + BT) * BK;          // 10240 floats = 40 KB
constexpr int kW8TileFloats = W8_I * BT;                  // partial slot: 64 KB
constexpr int kW8LdsBytes = kW8Stages * kW8StageFloats * 4;

struct Frag4 {
  float4 a0[4], a1[4], b0[4], b1[4];
};
__device__ __forceinline__ float f4c(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

__global__ __launch_bounds__(kBlock, 1) void gemm_w8_kernel(const float* __restrict__ A, int64_t lda,
                                                            const float* __restrict__ B, int64_t ldb, int64_t I, int64_t J,
                                                            int tiles_j, int S, int64_t total_units, int upb,
                                                            float* __restrict__ C, int64_t ldc, int transpose_out,
                                                            const float* __restrict__ bias,
                                                            const uint8_t* __restrict__ keep, float scale,
                                                            float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float ring[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, lr = lane & 31;
  const int sw = (lr >> 1) & 7;
  const int ia0 = (w * 64 + lr) * BK, ia1 = ia0 + 32 * BK;
  const int jb0 = W8_I * BK + lr * BK, jb1 = jb0 + 32 * BK;
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  const int64_t u_begin = (int64_t)blockIdx.x * upb;
  const int64_t u_end = min(total_units, u_begin + upb);
  int64_t u = u_begin;
  while (u < u_end) {
    const int64_t tile = u / S;
    const int s0 = (int)(u - tile * S);
    const int s1 = (int)min((int64_t)S, s0 + (u_end - u));
    const int nk = s1 - s0;
    const int64_t i0 = (tile / tiles_j) * W8_I, j0 = (tile % tiles_j) * BT;
    const float* pa[8];
    const float* pb[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = 64 * w + 8 * j + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      pa[j] = A + min(i0 + r, I - 1) * lda + (int64_t)s0 * BK + 4 * c;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 16 * w + 8 * j + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      pb[j] = B + min(j0 + r, J - 1) * ldb + (int64_t)s0 * BK + 4 * c;
    }
    const unsigned dst_a = __builtin_amdgcn_readfirstlane((unsigned)(64 * w * BK * 4));
    const unsigned dst_b = __builtin_amdgcn_readfirstlane((unsigned)((W8_I * BK + 16 * w * BK) * 4));
    // one of the 10 DMA instructions of slice kt (p < 8: my A rows, else my B rows)
    auto issue_one = [&](int kt, int p) {
      const unsigned st = ring_lds + (unsigned)(kt % kW8Stages) * (kW8StageFloats * 4);
      if (p < 8) glds16(pa[p] + (int64_t)kt * BK, st + dst_a + p * 1024);
      else glds16(pb[p - 8] + (int64_t)kt * BK, st + dst_b + (p - 8) * 1024);
    };
    // one of the 16 fragment float4s of slice kt
    auto read_one = [&](int kt, int g, Frag4& f) {
      const float* st = ring + (kt % kW8Stages) * kW8StageFloats;
      const int q = g & 3;
      const int pos = ((2 * q + h) ^ sw) * 4;
      if (g < 4) f.a0[q] = *reinterpret_cast<const float4*>(st + ia0 + pos);
      else if (g < 8) f.a1[q] = *reinterpret_cast<const float4*>(st + ia1 + pos);
      else if (g < 12) f.b0[q] = *reinterpret_cast<const float4*>(st + jb0 + pos);
      else f.b1[q] = *reinterpret_cast<const float4*>(st + jb1 + pos);
    };
    floatx16 c00, c01, c10, c11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { c00[r] = 0.f; c01[r] = 0.f; c10[r] = 0.f; c11[r] = 0.f; }
    // slice kt from `cur`; meanwhile DMA of slice kt+3 and the fragments of slice kt+1 into `nxt`.
    // MORE: slice kt+1 exists, FAR: slice kt+3 exists (compile-time, so the steady-state body is branch-free and the
    // accumulators stay in their AGPRs across iterations)
    auto body = [&](int kt, const Frag4& cur, Frag4& nxt, auto more_c, auto far_c, auto two_c) {
      constexpr bool MORE = decltype(more_c)::value, FAR = decltype(far_c)::value, TWO = decltype(two_c)::value;
      if (MORE) {
        if (TWO) vm_wait_n<10>();      // slice kt+2 is in flight behind slice kt+1
        else vm_wait_n<0>();
        lgkm_wait0();
        bare_barrier();                // slice kt+1 has landed everywhere; nobody still reads stage kt % 3
      }
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const int q = g >> 2, e = g & 3;
        const float a0 = f4c(cur.a0[q], e), a1 = f4c(cur.a1[q], e), b0 = f4c(cur.b0[q], e), b1 = f4c(cur.b1[q], e);
        c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c00, 0, 0, 0);
        c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c01, 0, 0, 0);
        c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c10, 0, 0, 0);
        c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c11, 0, 0, 0);
        if (FAR && g < 10) issue_one(kt + 3, g);
        if (MORE) read_one(kt + 1, g, nxt);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    const std::true_type T_;
    const std::false_type F_;
    vm_wait_n<0>();
    lgkm_wait0();
    bare_barrier();                    // ring hand-over from the previous range
    for (int t = 0; t < 3 && t < nk; ++t)
#pragma unroll
      for (int p = 0; p < 10; ++p) issue_one(t, p);
    if (nk > 2) vm_wait_n<20>();
    else if (nk > 1) vm_wait_n<10>();
    else vm_wait_n<0>();
    bare_barrier();
    Frag4 f0, f1;
#pragma unroll
    for (int g = 0; g < 16; ++g) read_one(0, g, f0);
    int kt = 0;
    for (; kt + 4 < nk; kt += 2) {             // steady state: slices kt+3 and kt+4 exist
      body(kt, f0, f1, T_, T_, T_);
      body(kt + 1, f1, f0, T_, T_, T_);
    }
    for (; kt < nk; kt += 2) {                 // drain (<= 4 slices): the same body with its conditions resolved
      const int left = nk - kt;                // slices kt .. nk-1
      if (left >= 4) body(kt, f0, f1, T_, T_, T_);
      else if (left == 3) body(kt, f0, f1, T_, F_, T_);
      else if (left == 2) body(kt, f0, f1, T_, F_, F_);
      else body(kt, f0, f1, F_, F_, F_);
      if (left >= 2) {
        if (left >= 5) body(kt + 1, f1, f0, T_, T_, T_);
        else if (left == 4) body(kt + 1, f1, f0, T_, F_, T_);
        else if (left == 3) body(kt + 1, f1, f0, T_, F_, F_);
        else body(kt + 1, f1, f0, F_, F_, F_);
      }
    }
    const bool whole = (s0 == 0 && s1 == S);
    const int seg = (u == u_begin) ? 0 : 1;
    float4* P = reinterpret_cast<float4*>(partials + ((size_t)blockIdx.x * 2 + seg) * kW8TileFloats);
#pragma unroll
    for (int ab = 0; ab < 4; ++ab) {
      const floatx16& acc = ab == 0 ? c00 : (ab == 1 ? c01 : (ab == 2 ? c10 : c11));
      const int a = ab >> 1, b = ab & 1;
      if (whole) {
        const int64_t col = j0 + 32 * b + lr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t row = i0 + w * 64 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (row < I && col < J) {
            float v = acc[r];
            if (bias) v += bias[col];
            if (keep) v = keep[row * J + col] ? v * scale : 0.f;
            if (transpose_out) C[col * ldc + row] = v;
            else C[row * ldc + col] = v;
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          P[(ab * 4 + q) * kBlock + tid] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
      }
    }
    u += nk;
  }
}

__global__ __launch_bounds__(kBlock) void w8_reduce_kernel(const float* __restrict__ partials, int tiles_j, int S,
                                                           int64_t total_units, int upb, int64_t I, int64_t J,
                                                           float* __restrict__ C, int64_t ldc, int transpose_out,
                                                           const float* __restrict__ bias,
                                                           const uint8_t* __restrict__ keep, float scale) {
  const int64_t tile = blockIdx.x;
  const int64_t u_lo = tile * S, u_hi = u_lo + S;
  const int64_t b_first = u_lo / upb, b_last = (u_hi - 1) / upb;
  if (b_first == b_last) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, lr = lane & 31;
  const int64_t i0 = (tile / tiles_j) * W8_I, j0 = (tile % tiles_j) * BT;
  for (int ab = 0; ab < 4; ++ab) {
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t b = b_first; b <= b_last; ++b) {
      const int seg = (b * upb < u_lo) ? 1 : 0;
      const float4* P = reinterpret_cast<const float4*>(partials + ((size_t)b * 2 + seg) * kW8TileFloats);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 p = P[(ab * 4 + q) * kBlock + tid];
        v[q].x += p.x; v[q].y += p.y; v[q].z += p.z; v[q].w += p.w;
      }
    }
    const int a = ab >> 1, bb = ab & 1;
    const int64_t col = j0 + 32 * bb + lr;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float e[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int r = 4 * q + c;
        const int64_t row = i0 + w * 64 + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < I && col < J) {
          float x = e[c];
          if (bias) x += bias[col];
          if (keep) x = keep[row * J + col] ? x * scale : 0.f;
          if (transpose_out) C[col * ldc + row] = x;
          else C[row * ldc + col] = x;
        }
      }
    }
  }
}

// ======================================================================================
// v9 (MMSSL_GEMM_V=9): register-direct product. No LDS and no barrier in the main loop: every wave loads its MFMA
// fragments straight from global memory with 16-byte loads and keeps two full fragment sets in registers.
//
// Why this is possible: v_mfma_f32_32x32x2_f32 wants lane l to hold A[row l%32][k-index l/32]; the sum over k does
// not care WHICH two k-values form a pair as long as A and B agree. So lane l reads 64 contiguous bytes of its row,
// k = kw + 16 (l/32) + 0..15, for both operands: MFMA step j multiplies the pair (kw + j, kw + 16 + j). A 32-row
// block then costs 4 x global_load_dwordx4 per 32 k-values and lanes l / l+32 cover one whole 128-B line of the row.
// Block = 4 waves on ONE 128 x 64 output tile; wave w takes k in [128 s + 32 w, +32) of every 128-deep unit s
// (the block reads 512 contiguous bytes of each row per unit), holds 4 x 2 accumulators (128 registers) and two
// fragment sets of 24 float4 (192 registers): 1 wave per SIMD, 0.75 loaded registers per MFMA (v6: 2, v7: 1.5, v8: 1),
// one load instruction per 5.3 MFMAs, issued between groups of four MFMAs on four different accumulators.
// The four waves' accumulators are added through LDS in wave order at the end of a unit range (fixed order);
// ranges are cut stream-K style like v6 (head / tail partial slots + r9_reduce_kernel in block order).
// Preconditions (host-checked): KK % 128 == 0, row-major [i][kk] operands, 16-B aligned rows.
// ======================================================================================
constexpr int R9_RB = 4;
constexpr int R9_I = 32 * R9_RB;                          // 128 rows per tile
constexpr int R9_KU = 128;                                // k-values per unit (4 waves x 32)
constexpr int kR9TileFloats = R9_I * BT;                  // partial slot: 32 KB
constexpr int kR9LdsBytes = 4 * kR9TileFloats * 4;        // the four waves' accumulator images: 128 KB

struct R9Frag {
  float4 a[R9_RB][4], b[2][4];
};
__device__ __forceinline__ float4 ldg16(const float* p) { return *reinterpret_cast<const float4*>(p); }

template <int DBG>      // DBG (debugging aid, MMSSL_GEMM_R9_MODE): 1 = no loads in the loop, 2 = no MFMAs, 4 = no B loads in the loop
__global__ __launch_bounds__(kBlock, 1) void gemm_r9_kernel(const float* __restrict__ A, int64_t lda,
                                                            const float* __restrict__ B, int64_t ldb, int64_t I,
                                                            int64_t J, int tiles_j, int S, int64_t total_units, int upb,
                                                            float* __restrict__ C, int64_t ldc, int transpose_out,
                                                            const float* __restrict__ bias,
                                                            const uint8_t* __restrict__ keep, float scale,
                                                            float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float4 r9_red[];        // [4 waves][2048]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, lr = lane & 31;
  const int64_t u_begin = (int64_t)blockIdx.x * upb;
  const int64_t u_end = min(total_units, u_begin + upb);
  int64_t u = u_begin;
  while (u < u_end) {
    const int64_t tile = u / S;
    const int s0 = (int)(u - tile * S);
    const int s1 = (int)min((int64_t)S, s0 + (u_end - u));
    const int nk = s1 - s0;
    const int64_t i0 = (tile / tiles_j) * R9_I, j0 = (tile % tiles_j) * BT;
    const float* pa[R9_RB];
    const float* pb[2];
    const int64_t k_first = (int64_t)s0 * R9_KU + 32 * wave + 16 * h;
#pragma unroll
    for (int rb = 0; rb < R9_RB; ++rb) pa[rb] = A + min(i0 + 32 * rb + lr, I - 1) * lda + k_first;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) pb[cb] = B + min(j0 + 32 * cb + lr, J - 1) * ldb + k_first;
    floatx16 acc[R9_RB][2];
#pragma unroll
    for (int rb = 0; rb < R9_RB; ++rb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[rb][0][r] = 0.f;
        acc[rb][1][r] = 0.f;
      }
    // one unit: 128 MFMAs from `cur`, the 24 loads of unit `kn` into `nxt` spread between them
    auto step = [&](int kn, const R9Frag& cur, R9Frag& nxt) {
      const int64_t off = (int64_t)kn * R9_KU;
#pragma unroll
      for (int i = 0; i < 16; ++i) {            // row blocks 0, 1; fetches a[0], b[0], b[1], a[1] of the next unit
        const int q = i >> 2, c = i & 3;
        const float x0 = f4c(cur.a[0][q], c), x1 = f4c(cur.a[1][q], c);
        const float y0 = f4c(cur.b[0][q], c), y1 = f4c(cur.b[1][q], c);
        if (DBG & 2) {
          asm volatile("" ::"v"(x0), "v"(x1), "v"(y0), "v"(y1));
        } else {
          acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc[0][0], 0, 0, 0);
          acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc[0][1], 0, 0, 0);
          acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc[1][0], 0, 0, 0);
          acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc[1][1], 0, 0, 0);
        }
        if (!(DBG & 1)) {
          if (c == 0) nxt.a[0][q] = ldg16(pa[0] + off + 4 * q);
          else if (c == 1 && !(DBG & 4)) nxt.b[0][q] = ldg16(pb[0] + off + 4 * q);
          else if (c == 2 && !(DBG & 4)) nxt.b[1][q] = ldg16(pb[1] + off + 4 * q);
          else if (c == 3) nxt.a[1][q] = ldg16(pa[1] + off + 4 * q);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) {            // row blocks 2, 3; fetches a[2], a[3]
        const int q = i >> 2, c = i & 3;
        const float x0 = f4c(cur.a[2][q], c), x1 = f4c(cur.a[3][q], c);
        const float y0 = f4c(cur.b[0][q], c), y1 = f4c(cur.b[1][q], c);
        if (DBG & 2) {
          asm volatile("" ::"v"(x0), "v"(x1), "v"(y0), "v"(y1));
        } else {
          acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc[2][0], 0, 0, 0);
          acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc[2][1], 0, 0, 0);
          acc[3][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc[3][0], 0, 0, 0);
          acc[3][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc[3][1], 0, 0, 0);
        }
        if (!(DBG & 1)) {
          if (c == 0) nxt.a[2][q] = ldg16(pa[2] + off + 4 * q);
          else if (c == 2) nxt.a[3][q] = ldg16(pa[3] + off + 4 * q);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    R9Frag f0, f1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f0.a[0][q] = ldg16(pa[0] + 4 * q);
      f0.b[0][q] = ldg16(pb[0] + 4 * q);
      f0.b[1][q] = ldg16(pb[1] + 4 * q);
      f0.a[1][q] = ldg16(pa[1] + 4 * q);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f0.a[2][q] = ldg16(pa[2] + 4 * q);
      f0.a[3][q] = ldg16(pa[3] + 4 * q);
    }
    if (DBG & 5) f1 = f0;
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {              // the prefetch of the last unit re-reads that unit (never used)
      step(kt + 1, f0, f1);
      step(min(kt + 2, nk - 1), f1, f0);
    }
    if (kt < nk) step(kt, f0, f1);
    // add the four waves' accumulators in wave order: wave w finishes row block w
    float4* mine = r9_red + wave * (kR9TileFloats / 4);
#pragma unroll
    for (int rb = 0; rb < R9_RB; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          mine[((rb * 2 + cb) * 4 + q) * 64 + lane] =
              make_float4(acc[rb][cb][4 * q], acc[rb][cb][4 * q + 1], acc[rb][cb][4 * q + 2], acc[rb][cb][4 * q + 3]);
    __syncthreads();
    const bool whole = (s0 == 0 && s1 == S);
    const int seg = (u == u_begin) ? 0 : 1;
    float4* P = reinterpret_cast<float4*>(partials + ((size_t)blockIdx.x * 2 + seg) * kR9TileFloats);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = ((wave * 2 + cb) * 4 + q) * 64 + lane;
        const float4 r0 = r9_red[idx], r1 = r9_red[2048 + idx], r2 = r9_red[4096 + idx], r3 = r9_red[6144 + idx];
        float4 v;
        v.x = ((r0.x + r1.x) + r2.x) + r3.x;
        v.y = ((r0.y + r1.y) + r2.y) + r3.y;
        v.z = ((r0.z + r1.z) + r2.z) + r3.z;
        v.w = ((r0.w + r1.w) + r2.w) + r3.w;
        if (!whole) {
          P[idx] = v;
        } else {
          const int64_t col = j0 + 32 * cb + lr;
          const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int r = 4 * q + c;
            const int64_t row = i0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (row < I && col < J) {
              float x = e[c];
              if (bias) x += bias[col];
              if (keep) x = keep[row * J + col] ? x * scale : 0.f;
              if (transpose_out) C[col * ldc + row] = x;
              else C[row * ldc + col] = x;
            }
          }
        }
      }
    __syncthreads();                            // the images are reused by the block's next range
    u += nk;
  }
}

// one block per 128x64 output tile: adds the tile's partial slots in block order
__global__ __launch_bounds__(kBlock) void r9_reduce_kernel(const float* __restrict__ partials, int tiles_j, int S,
                                                           int64_t total_units, int upb, int64_t I, int64_t J,
                                                           float* __restrict__ C, int64_t ldc, int transpose_out,
                                                           const float* __restrict__ bias,
                                                           const uint8_t* __restrict__ keep, float scale) {
  const int64_t tile = blockIdx.x;
  const int64_t u_lo = tile * S, u_hi = u_lo + S;
  const int64_t b_first = u_lo / upb, b_last = (u_hi - 1) / upb;
  if (b_first == b_last) return;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int h = lane >> 5, lr = lane & 31;
  float4 v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t b = b_first; b <= b_last; ++b) {
    const int seg = (b * upb < u_lo) ? 1 : 0;
    const float4* P = reinterpret_cast<const float4*>(partials + ((size_t)b * 2 + seg) * kR9TileFloats);
    float4 p[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) p[q] = P[(w * 8 + q) * 64 + lane];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      v[q].x += p[q].x; v[q].y += p[q].y; v[q].z += p[q].z; v[q].w += p[q].w;
    }
  }
  const int64_t i0 = (tile / tiles_j) * R9_I, j0 = (tile % tiles_j) * BT;
#pragma unroll
  for (int q8 = 0; q8 < 8; ++q8) {
    const int cb = q8 >> 2, q = q8 & 3;
    const int64_t col = j0 + 32 * cb + lr;
    const float e[4] = {v[q8].x, v[q8].y, v[q8].z, v[q8].w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r = 4 * q + c;
      const int64_t row = i0 + w * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < I && col < J) {
        float x = e[c];
        if (bias) x += bias[col];
        if (keep) x = keep[row * J + col] ? x * scale : 0.f;
        if (transpose_out) C[col * ldc + row] = x;
        else C[row * ldc + col] = x;
      }
    }
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
  // Register ring of WG_D steps. The loads are inline asm so that neither the IR passes nor the machine scheduler can
  // sink them next to their use (both did, which serialised load latency and MFMAs); the matching waits therefore are
  // explicit: when slot d is consumed, exactly WG_D - 1 younger steps of loads are outstanding.
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
  constexpr int kYounger = (KEEP ? 3 : 2) * (WG_D - 1);
  auto load = [&](int slot, int t) {
    const uint32_t tc = (uint32_t)min(t, t_last);
    const uint32_t a = oF + __umul24(tc, sF), b = oG + __umul24(tc, sG);
    if (NT) asm volatile("global_load_dwordx4 %0, %1, %2 nt" : "=v"(fv[slot]) : "v"(a), "s"(F) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(fv[slot]) : "v"(a), "s"(F) : "memory");
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(gv[slot]) : "v"(b), "s"(gY) : "memory");
    if (KEEP) {
      const uint32_t c = oK + __umul24(tc, sK);
      asm volatile("global_load_dword %0, %1, %2" : "=v"(kv[slot]) : "v"(c), "s"(keep) : "memory");
    }
  };
  auto consume = [&](int slot, int t) {
    if (KEEP) asm volatile("s_waitcnt vmcnt(%3)" : "+v"(fv[slot]), "+v"(gv[slot]), "+v"(kv[slot]) : "n"(kYounger));
    else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(fv[slot]), "+v"(gv[slot]) : "n"(kYounger));
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
    // pin: the MFMAs of this step are issued HERE (left alone, the DAG scheduler defers them past the next loads)
#pragma unroll
    for (int rb = 0; rb < 4; rb += 2)
      asm volatile("" : "+v"(acc[rb][0]), "+v"(acc[rb][1]), "+v"(acc[rb][2]), "+v"(acc[rb][3]), "+v"(acc[rb + 1][0]),
                        "+v"(acc[rb + 1][1]), "+v"(acc[rb + 1][2]), "+v"(acc[rb + 1][3]));
  };
#pragma unroll
  for (int d = 0; d < WG_D; ++d) load(d, d);
  for (int t0 = 0; t0 < nsteps; t0 += WG_D) {
#pragma unroll
    for (int d = 0; d < WG_D; ++d) {
      consume(d, t0 + d);
      __builtin_amdgcn_sched_barrier(0);
      load(d, t0 + d + WG_D);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the ring's last prefetches still target live registers
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

// ======================================================================================
// v10 forward over a transposed copy of the feature matrix (opt-in, mmssl_linear_ft_f32):
//   Y^T [N, Mp] = sum_k WT[k][n] * FT[k][m]  is the weight-gradient form (reduction over the rows of two row-major
// operands), so the same register-direct kernel streams FT with fully coalesced 16-byte loads. W (64 x K, 1 MB) is
// transposed per call by w_transpose_kernel; the row-range partials are added, transposed back through LDS and get bias
// + dropout in ft_reduce_kernel.
// ======================================================================================
__global__ __launch_bounds__(kBlock) void w_transpose_kernel(const float* __restrict__ W, int N, int K,
                                                             float* __restrict__ WT) {
  __shared__ float tile[64][65];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int k0 = (int)blockIdx.x * 64, n0 = (int)blockIdx.y * 64;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int n = q + 4 * i;
    tile[n][c] = (n0 + n < N && k0 + c < K) ? W[(int64_t)(n0 + n) * K + k0 + c] : 0.f;
  }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int k = q + 4 * i;
    if (k0 + k < K && n0 + c < N) WT[(int64_t)(k0 + k) * N + n0 + c] = tile[c][k];
  }
}

// one block per 64 output rows (m) x 64 columns (n): Y[m][n] = dropout(sum_s P[s][n][m] + bias[n])
__global__ __launch_bounds__(kBlock) void ft_reduce_kernel(const float* __restrict__ P, int sp, int64_t split_stride,
                                                           int64_t Mp, int64_t M, int N, float* __restrict__ Y,
                                                           const float* __restrict__ bias,
                                                           const uint8_t* __restrict__ keep, float scale) {
  __shared__ float tile[64][65];
  const int t = threadIdx.x;
  const int64_t m0 = (int64_t)blockIdx.x * 64;
  const int n0 = (int)blockIdx.y * 64;
  {
    const int n = t >> 2, mc = (t & 3) * 16;
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < sp; ++s) {
      const float4* src = reinterpret_cast<const float4*>(P + (size_t)s * split_stride + (int64_t)(n0 + n) * Mp + m0 + mc);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 p = src[q];
        v[q].x += p.x; v[q].y += p.y; v[q].z += p.z; v[q].w += p.w;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      tile[n][mc + 4 * q + 0] = v[q].x;
      tile[n][mc + 4 * q + 1] = v[q].y;
      tile[n][mc + 4 * q + 2] = v[q].z;
      tile[n][mc + 4 * q + 3] = v[q].w;
    }
  }
  __syncthreads();
  const int m = t >> 2, nc = (t & 3) * 16;
  if (m0 + m >= M) return;
  float o[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float x = tile[nc + i][m];
    if (bias) x += bias[n0 + nc + i];
    o[i] = x;
  }
  if (keep) {
    const uint4 kk = *reinterpret_cast<const uint4*>(keep + (m0 + m) * N + n0 + nc);
    const unsigned kw[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = ((kw[i >> 2] >> (8 * (i & 3))) & 0xffu) ? o[i] * scale : 0.f;
  }
  float4* dst = reinterpret_cast<float4*>(Y + (m0 + m) * N + n0 + nc);
#pragma unroll
  for (int q = 0; q < 4; ++q) dst[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}

// ======================================================================================
// OPT-IN split-precision product (MMSSL_GEMM_SPLIT=1 on the Python side; NOT the default path):
//   C[i][j] = sum_k A[i][k] * B[j][k]   with A, B given as bf16 (hi, lo) pairs, x ~= hi + lo (16 mantissa bits),
//   accumulated in fp32 as  hi*hi + hi*lo + lo*hi  on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16).
// Why it exists: bf16 matrix instructions have 8x the fp32 rate (and cost ~0.13x the energy per flop,
// profiles/r01_power_probe.txt), so three bf16 products are cheaper than one fp32 product; the operand pairs occupy the
// same bytes as fp32. (Round 1 argued the fp32 kernel was held back by the power cap; round 2 measured that it is not -
// DESIGN.md section 4 - so this path is an arithmetic trade, not a power workaround.)
// Same structure as gemm_fwd_dma_kernel: 64x64 block tile, 2x2 waves, 32-deep slices, 4-stage LDS-DMA ring.
// One stage = four 4 KB tiles (A_hi, A_lo, B_hi, B_lo), each 64 rows x 32 bf16 = 4 chunks of 16 B per row;
// chunk c of row r sits at slot 4r + (c ^ ((r >> 2) & 3)) (swizzle applied to the DMA source address), which makes
// the 16-lane groups of ds_read_b128 conflict-free with the 64-B row pitch.
// ======================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kSplitTileBytes = BT * BK * 2;                 // 4 KB
constexpr int kSplitStageBytes = 4 * kSplitTileBytes;        // 16 KB

__global__ __launch_bounds__(kBlock) void gemm_split_kernel(const uint16_t* __restrict__ Ahi,
                                                            const uint16_t* __restrict__ Alo, int64_t lda,
                                                            const uint16_t* __restrict__ Bhi,
                                                            const uint16_t* __restrict__ Blo, int64_t ldb, int64_t I,
                                                            int64_t J, int64_t kk_chunk, float* __restrict__ C,
                                                            int64_t ldc, int64_t split_stride,
                                                            const float* __restrict__ bias,
                                                            const uint8_t* __restrict__ keep, float scale) {
  __shared__ __attribute__((aligned(16))) unsigned char ring[kDmaStages * kSplitStageBytes];   // 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t i0 = (int64_t)blockIdx.x * BT, j0 = (int64_t)blockIdx.y * BT;
  const int64_t kk_beg = (int64_t)blockIdx.z * kk_chunk;
  const int nk = (int)(kk_chunk / BK);
  floatx16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // DMA pieces: wave w moves rows [16w, 16w+16) of each of the four tiles (16 rows x 64 B = 1 KiB per piece)
  const int pr = 16 * wave + (lane >> 2);
  const int pc = (lane & 3) ^ ((pr >> 2) & 3);                 // source-side swizzle
  const int64_t ra = min(i0 + pr, I - 1) * lda + kk_beg + 8 * pc;
  const int64_t rb = min(j0 + pr, J - 1) * ldb + kk_beg + 8 * pc;
  const uint16_t* src[4] = {Ahi + ra, Alo + ra, Bhi + rb, Blo + rb};
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  const unsigned piece = __builtin_amdgcn_readfirstlane((unsigned)(16 * wave * BK * 2));
  auto issue = [&](int kt) {
    const unsigned st = ring_lds + (unsigned)(kt & (kDmaStages - 1)) * kSplitStageBytes + piece;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      glds16(reinterpret_cast<const float*>(src[t] + (int64_t)kt * BK), st + t * kSplitTileBytes);
  };
  issue(0);
  issue(1);
  issue(2);
  const int h = lane >> 5, lr = lane & 31;
  const int ia = (wm * 32 + lr) * (BK * 2), jb = (wn * 32 + lr) * (BK * 2);      // byte offsets of the lane's rows
  const int swa = ((wm * 32 + lr) >> 2) & 3, swb = ((wn * 32 + lr) >> 2) & 3;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 2 < nk) vm_wait_n<8>();
    else if (kt + 1 < nk) vm_wait_n<4>();
    else vm_wait_n<0>();
    bare_barrier();
    if (kt + 3 < nk) issue(kt + 3);
    const unsigned char* st = ring + (kt & (kDmaStages - 1)) * kSplitStageBytes;
#pragma unroll
    for (int t = 0; t < 2; ++t) {                // two 16-deep MFMA steps per slice; lane half h owns k = 16t + 8h .. +7
      const int ca = ((2 * t + h) ^ swa) * 16, cb = ((2 * t + h) ^ swb) * 16;
      const bf16x8 ah = *reinterpret_cast<const bf16x8*>(st + ia + ca);
      const bf16x8 al = *reinterpret_cast<const bf16x8*>(st + kSplitTileBytes + ia + ca);
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(st + 2 * kSplitTileBytes + jb + cb);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(st + 3 * kSplitTileBytes + jb + cb);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);      // small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
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

// ---- operand preparation for the opt-in split-precision product -------------------------------------------
__device__ __forceinline__ uint16_t bf16_rne(float x) {        // round-to-nearest-even, finite inputs
  unsigned b = __float_as_uint(x);
  b += 0x7fffu + ((b >> 16) & 1u);
  return (uint16_t)(b >> 16);
}
__device__ __forceinline__ void bf16_pair(float x, uint16_t& hi, uint16_t& lo) {
  hi = bf16_rne(x);
  lo = bf16_rne(x - __uint_as_float((unsigned)hi << 16));
}

__global__ __launch_bounds__(kBlock) void split_pair_kernel(const float4* __restrict__ X, int64_t n4,
                                                            ushort4* __restrict__ hi, ushort4* __restrict__ lo) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    const float4 v = X[i];
    ushort4 h, l;
    bf16_pair(v.x, h.x, l.x);
    bf16_pair(v.y, h.y, l.y);
    bf16_pair(v.z, h.z, l.z);
    bf16_pair(v.w, h.w, l.w);
    hi[i] = h;
    lo[i] = l;
  }
}

// G [M, N] (optionally dropout-masked) -> T_hi, T_lo [N, Mp] bf16, transposed, zero past row M, plus per-block
// column sums of the masked G (-> bias gradient). One block = 64 rows x 64 columns through a padded LDS tile.
__global__ __launch_bounds__(kBlock) void split_transpose_kernel(const float* __restrict__ G,
                                                                 const uint8_t* __restrict__ keep, float scale,
                                                                 int64_t M, int N, int64_t Mp,
                                                                 uint16_t* __restrict__ Thi,
                                                                 uint16_t* __restrict__ Tlo,
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
    if (n0 + n < N) {
      uint16_t h, l;
      bf16_pair(tile[c][n], h, l);
      Thi[(int64_t)(n0 + n) * Mp + m0 + c] = h;
      Tlo[(int64_t)(n0 + n) * Mp + m0 + c] = l;
    }
  }
}

// Forward kernel choice. Default: the LDS-DMA kernel whenever the K range splits into whole 32-deep slices
// (every Tiktok / Baby / stress shape); MMSSL_GEMM_V=1 forces the register-staged kernel everywhere.
// Measured on MI355X, Baby image projection [18357,4096]x[4096,64], kernel + split reduce under a hipGraph:
// register-staged 137-141 us (70 TF), LDS-DMA 130-132 us (73-74 TF); DESIGN.md section 4 has the
// decomposition runs that explain why both sit near 75 TF.
inline int gemm_version() {
  static const int v = getenv("MMSSL_GEMM_V") ? atoi(getenv("MMSSL_GEMM_V")) : 6;
  return v;
}
inline bool dma_enabled() { return gemm_version() != 1; }
inline bool gemm_nt() {
  static const int v = getenv("MMSSL_GEMM_NT") ? atoi(getenv("MMSSL_GEMM_NT")) : 0;
  return v != 0;
}

// stream-K decomposition (gemm_sk_kernel): `slots` equal unit ranges, at least min(S, 8) slices each
struct SkPlan {
  int64_t tiles_i, tiles_j, total;
  int S, upb, blocks;
};
inline int sk_slots() {
  static const int v = getenv("MMSSL_GEMM_SK_BLOCKS") ? atoi(getenv("MMSSL_GEMM_SK_BLOCKS")) : 512;   // 2 per CU
  return v > 0 ? v : 512;
}
inline SkPlan sk_plan(int64_t I, int64_t J, int64_t KK) {
  SkPlan p;
  p.tiles_i = (I + BT - 1) / BT;
  p.tiles_j = (J + BT - 1) / BT;
  p.S = (int)(KK / BK);
  p.total = p.tiles_i * p.tiles_j * p.S;
  int64_t upb = (p.total + sk_slots() - 1) / sk_slots();
  const int64_t floor_ = p.S < 8 ? p.S : 8;
  if (upb < floor_) upb = floor_;
  p.upb = (int)upb;
  p.blocks = (int)((p.total + upb - 1) / upb);
  return p;
}
inline bool sk_usable(int64_t KK) { return gemm_version() >= 6 && KK % BK == 0 && KK >= BK; }

// v7 decomposition: 128x64 tiles, one 8-wave block per CU
inline int pp_slots() {
  static const int v = getenv("MMSSL_GEMM_PP_BLOCKS") ? atoi(getenv("MMSSL_GEMM_PP_BLOCKS")) : 256;
  return v > 0 ? v : 256;
}
inline int pp_bk() {
  static const int v = getenv("MMSSL_GEMM_PP_BK") ? atoi(getenv("MMSSL_GEMM_PP_BK")) : 16;
  return v == 32 ? 32 : 16;
}
inline SkPlan pp_plan(int64_t I, int64_t J, int64_t KK) {
  SkPlan p;
  p.tiles_i = (I + PT_I - 1) / PT_I;
  p.tiles_j = (J + BT - 1) / BT;
  p.S = (int)(KK / pp_bk());
  p.total = p.tiles_i * p.tiles_j * p.S;
  int64_t upb = (p.total + pp_slots() - 1) / pp_slots();
  const int64_t floor_ = p.S < 16 ? p.S : 16;
  if (upb < floor_) upb = floor_;
  p.upb = (int)upb;
  p.blocks = (int)((p.total + upb - 1) / upb);
  return p;
}
inline bool pp_usable(int64_t KK) { return gemm_version() == 7 && KK % BK == 0 && KK >= BK; }
constexpr int kPpLdsBytes = 144 * 1024;                  // 12 x 12 KB (PBK 16) or 6 x 24 KB (PBK 32)
inline int pp_lds_ready() {       // 144 KB of dynamic LDS needs the opt-in attribute once per instantiation
  static const int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp_kernel<16, 12, 10>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kPpLdsBytes) |
                        (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp_kernel<32, 6, 4>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kPpLdsBytes);
  return rc;
}
// C[I,J] (or its transpose) = A[I,KK] . B[J,KK]^T (+ bias, dropout) on the ping-pong kernel
inline int launch_pp(const float* A, const float* B, int64_t I, int64_t J, int64_t KK, float* C, int64_t ldc,
                     int transpose_out, const float* b, const uint8_t* keep, float scale, float* part, hipStream_t s) {
  if (pp_lds_ready() != 0) return MMSSL_E_UNSUPP;
  const SkPlan p = pp_plan(I, J, KK);
  const char* dm = getenv("MMSSL_GEMM_PP_MODE");        // debugging aid (tools/gemm_mode_probe.py)
  const int dbg = dm ? atoi(dm) : 0;
  if (pp_bk() == 16)
    hipLaunchKernelGGL((gemm_pp_kernel<16, 12, 10>), dim3((unsigned)p.blocks), dim3(kPpThreads), kPpLdsBytes, s, A, KK, B,
                       KK, I, J, (int)p.tiles_j, p.S, p.total, p.upb, C, ldc, transpose_out, b, keep, scale, part, dbg);
  else
    hipLaunchKernelGGL((gemm_pp_kernel<32, 6, 4>), dim3((unsigned)p.blocks), dim3(kPpThreads), kPpLdsBytes, s, A, KK, B,
                       KK, I, J, (int)p.tiles_j, p.S, p.total, p.upb, C, ldc, transpose_out, b, keep, scale, part, dbg);
  MMSSL_LAUNCH_CHECK();
  if (p.upb % p.S != 0) {
    hipLaunchKernelGGL(pp_reduce_kernel, dim3((unsigned)(p.tiles_i * p.tiles_j)), dim3(kBlock), 0, s, part,
                       (int)p.tiles_j, p.S, p.total, p.upb, I, J, C, ldc, transpose_out, b, keep, scale);
    MMSSL_LAUNCH_CHECK();
  }
  return 0;
}
inline bool w8_usable(int64_t KK) { return gemm_version() == 8 && KK % BK == 0 && KK >= BK; }
inline SkPlan w8_plan(int64_t I, int64_t J, int64_t KK) {
  SkPlan p;
  p.tiles_i = (I + W8_I - 1) / W8_I;
  p.tiles_j = (J + BT - 1) / BT;
  p.S = (int)(KK / BK);
  p.total = p.tiles_i * p.tiles_j * p.S;
  int64_t upb = (p.total + 255) / 256;
  const int64_t floor_ = p.S < 8 ? p.S : 8;
  if (upb < floor_) upb = floor_;
  p.upb = (int)upb;
  p.blocks = (int)((p.total + upb - 1) / upb);
  return p;
}
inline int launch_w8(const float* A, const float* B, int64_t I, int64_t J, int64_t KK, float* C, int64_t ldc,
                     int transpose_out, const float* b, const uint8_t* keep, float scale, float* part, hipStream_t s) {
  static const int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w8_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kW8LdsBytes);
  if (rc != 0) return MMSSL_E_UNSUPP;
  const SkPlan p = w8_plan(I, J, KK);
  hipLaunchKernelGGL(gemm_w8_kernel, dim3((unsigned)p.blocks), dim3(kBlock), kW8LdsBytes, s, A, KK, B, KK, I, J,
                     (int)p.tiles_j, p.S, p.total, p.upb, C, ldc, transpose_out, b, keep, scale, part);
  MMSSL_LAUNCH_CHECK();
  if (p.upb % p.S != 0) {
    hipLaunchKernelGGL(w8_reduce_kernel, dim3((unsigned)(p.tiles_i * p.tiles_j)), dim3(kBlock), 0, s, part,
                       (int)p.tiles_j, p.S, p.total, p.upb, I, J, C, ldc, transpose_out, b, keep, scale);
    MMSSL_LAUNCH_CHECK();
  }
  return 0;
}
inline bool r9_usable(int64_t KK) { return gemm_version() == 9 && KK % R9_KU == 0 && KK >= R9_KU; }
inline int r9_slots() {
  static const int v = getenv("MMSSL_GEMM_R9_BLOCKS") ? atoi(getenv("MMSSL_GEMM_R9_BLOCKS")) : 256;   // 1 per CU
  return v > 0 ? v : 256;
}
inline SkPlan r9_plan(int64_t I, int64_t J, int64_t KK) {
  SkPlan p;
  p.tiles_i = (I + R9_I - 1) / R9_I;
  p.tiles_j = (J + BT - 1) / BT;
  p.S = (int)(KK / R9_KU);
  p.total = p.tiles_i * p.tiles_j * p.S;
  int64_t upb = (p.total + r9_slots() - 1) / r9_slots();
  const int64_t floor_ = p.S < 2 ? p.S : 2;
  if (upb < floor_) upb = floor_;
  p.upb = (int)upb;
  p.blocks = (int)((p.total + upb - 1) / upb);
  return p;
}
inline int launch_r9(const float* A, const float* B, int64_t I, int64_t J, int64_t KK, float* C, int64_t ldc,
                     int transpose_out, const float* b, const uint8_t* keep, float scale, float* part, hipStream_t s) {
  static const int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_r9_kernel<0>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kR9LdsBytes) |
                        (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_r9_kernel<1>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kR9LdsBytes) |
                        (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_r9_kernel<2>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kR9LdsBytes) |
                        (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_r9_kernel<4>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kR9LdsBytes) |
                        (int)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_r9_kernel<6>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kR9LdsBytes);
  if (rc != 0) return MMSSL_E_UNSUPP;
  const SkPlan p = r9_plan(I, J, KK);
  const char* dm = getenv("MMSSL_GEMM_R9_MODE");        // debugging aid: 1 = no loads in the loop, 2 = no MFMAs
  const int dbg = dm ? atoi(dm) : 0;
  auto* kern = dbg == 1 ? gemm_r9_kernel<1> : (dbg == 2 ? gemm_r9_kernel<2> : (dbg == 4 ? gemm_r9_kernel<4> : (dbg == 6 ? gemm_r9_kernel<6> : gemm_r9_kernel<0>)));
  hipLaunchKernelGGL(kern, dim3((unsigned)p.blocks), dim3(kBlock), kR9LdsBytes, s, A, KK, B, KK, I, J,
                     (int)p.tiles_j, p.S, p.total, p.upb, C, ldc, transpose_out, b, keep, scale, part);
  MMSSL_LAUNCH_CHECK();
  if (p.upb % p.S != 0) {
    hipLaunchKernelGGL(r9_reduce_kernel, dim3((unsigned)(p.tiles_i * p.tiles_j)), dim3(kBlock), 0, s, part,
                       (int)p.tiles_j, p.S, p.total, p.upb, I, J, C, ldc, transpose_out, b, keep, scale);
    MMSSL_LAUNCH_CHECK();
  }
  return 0;
}
inline size_t pp_ws_bytes(int64_t I, int64_t J, int64_t KK) {
  return (size_t)pp_plan(I, J, KK).blocks * 2 * kPpTileFloats * sizeof(float) + 16;
}
// the ping-pong tile is 128 (A rows) x 64 (B rows): put the LONG operand on the A side, transposing the store
inline bool pp_swap(int64_t M, int64_t N) { return N > M; }

// split count: aim for >= ~4 blocks per CU, every split at least 4 slices deep
inline int choose_splits(int64_t tiles, int64_t KK) {
  const int64_t slices = (KK + BK - 1) / BK;
  if (const char* e = getenv("MMSSL_GEMM_SPLITS")) {      // tuning override (tools/gemm_sweep.py)
    const int64_t f = atoi(e);
    if (f >= 1) return (int)(f > slices ? slices : f);
  }
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
  if (r9_usable(K)) {
    const SkPlan p = pp_swap(M, N) ? r9_plan(N, M, K) : r9_plan(M, N, K);
    return (size_t)p.blocks * 2 * kR9TileFloats * sizeof(float) + 16;
  }
  if (w8_usable(K)) {
    const SkPlan p = pp_swap(M, N) ? w8_plan(N, M, K) : w8_plan(M, N, K);
    return (size_t)p.blocks * 2 * kW8TileFloats * sizeof(float) + 16;
  }
  if (pp_usable(K)) return pp_swap(M, N) ? pp_ws_bytes(N, M, K) : pp_ws_bytes(M, N, K);
  if (sk_usable(K)) return (size_t)sk_plan(M, N, K).blocks * 2 * kSkTileFloats * sizeof(float) + 16;
  const int64_t tiles = ((M + BT - 1) / BT) * ((N + BT - 1) / BT);
  const int splits = choose_splits(tiles, K);
  return splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 16;
}

static int linear_impl(const float* F, const float* W, const float* b, const uint8_t* keep, float scale, int64_t M,
                       int K, int N, float* Y, void* workspace, size_t workspace_bytes, int* tickets, void* stream);

extern "C" int mmssl_linear_f32(const float* F, const float* W, const float* b, const uint8_t* keep, float scale,
                                int64_t M, int K, int N, float* Y, void* workspace, size_t workspace_bytes,
                                void* stream) {
  return linear_impl(F, W, b, keep, scale, M, K, N, Y, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int64_t mmssl_linear_ticket_count(int64_t M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0 || gemm_version() != 6 || !sk_usable(K)) return 0;
  const SkPlan p = sk_plan(M, N, K);
  return p.tiles_i * p.tiles_j;
}

extern "C" int mmssl_linear_tk_f32(const float* F, const float* W, const float* b, const uint8_t* keep, float scale,
                                   int64_t M, int K, int N, float* Y, void* workspace, size_t workspace_bytes,
                                   int* tickets, void* stream) {
  if (tickets && (((uintptr_t)tickets & 3) || mmssl_linear_ticket_count(M, K, N) == 0)) return MMSSL_E_BADARG;
  return linear_impl(F, W, b, keep, scale, M, K, N, Y, workspace, workspace_bytes, tickets, stream);
}

static int linear_impl(const float* F, const float* W, const float* b, const uint8_t* keep, float scale, int64_t M,
                       int K, int N, float* Y, void* workspace, size_t workspace_bytes, int* tickets, void* stream) {
  if (M < 0 || K <= 0 || N <= 0 || (M > 0 && (!F || !W || !Y))) return MMSSL_E_BADARG;
  if ((K & 3) || (N & 3)) return MMSSL_E_UNSUPP;
  if (N > 256 && (b || keep || !sk_usable(K))) return MMSSL_E_UNSUPP;   // wide outputs: plain product only (wgrad)
  if (M == 0) return 0;
  if (((uintptr_t)F | (uintptr_t)W | (uintptr_t)Y) & 15) return MMSSL_E_BADARG;
  hipStream_t s = as_stream(stream);
  if (r9_usable(K)) {
    if (!workspace || workspace_bytes < mmssl_linear_workspace_bytes(M, K, N)) return MMSSL_E_WORKSPACE;
    float* part = reinterpret_cast<float*>(workspace);
    if (pp_swap(M, N)) return launch_r9(W, F, N, M, K, Y, (int64_t)N, 1, nullptr, nullptr, 1.f, part, s);
    return launch_r9(F, W, M, N, K, Y, (int64_t)N, 0, b, keep, scale, part, s);
  }
  if (w8_usable(K)) {
    if (!workspace || workspace_bytes < mmssl_linear_workspace_bytes(M, K, N)) return MMSSL_E_WORKSPACE;
    float* part = reinterpret_cast<float*>(workspace);
    if (pp_swap(M, N)) return launch_w8(W, F, N, M, K, Y, (int64_t)N, 1, nullptr, nullptr, 1.f, part, s);
    return launch_w8(F, W, M, N, K, Y, (int64_t)N, 0, b, keep, scale, part, s);
  }
  if (pp_usable(K)) {
    if (!workspace || workspace_bytes < mmssl_linear_workspace_bytes(M, K, N)) return MMSSL_E_WORKSPACE;
    float* part = reinterpret_cast<float*>(workspace);
    // Y[M,N] = F[M,K] . W[N,K]^T. Wide outputs (the weight gradient: M = 64, N = 4096) run as the transposed product
    // W . F^T with a transposed store, so that the 128-row side of the tile is the long one.
    if (pp_swap(M, N)) return launch_pp(W, F, N, M, K, Y, (int64_t)N, 1, nullptr, nullptr, 1.f, part, s);
    return launch_pp(F, W, M, N, K, Y, (int64_t)N, 0, b, keep, scale, part, s);
  }
  if (sk_usable(K)) {
    const SkPlan p = sk_plan(M, N, K);
    if (!workspace || workspace_bytes < (size_t)p.blocks * 2 * kSkTileFloats * sizeof(float)) return MMSSL_E_WORKSPACE;
    float* part = reinterpret_cast<float*>(workspace);
    const char* pr = getenv("MMSSL_GEMM_PRIO");
    hipLaunchKernelGGL(gemm_sk_kernel, dim3((unsigned)p.blocks), dim3(kBlock), 0, s, F, (int64_t)K, W, (int64_t)K, M,
                       (int64_t)N, (int)p.tiles_j, p.S, p.total, p.upb, Y, (int64_t)N, b, keep, scale, part,
                       ((pr ? atoi(pr) : 0) & 1) | (gemm_nt() ? 2 : 0), tickets);
    MMSSL_LAUNCH_CHECK();
    if (p.upb % p.S != 0 && !tickets) {        // some range ends inside a tile: partial slots exist
      hipLaunchKernelGGL(sk_reduce_kernel, dim3((unsigned)(p.tiles_i * p.tiles_j)), dim3(kBlock), 0, s, part,
                         (int)p.tiles_j, p.S, p.total, p.upb, M, (int64_t)N, Y, (int64_t)N, b, keep, scale);
      MMSSL_LAUNCH_CHECK();
    }
    return 0;
  }
  const int64_t tm = (M + BT - 1) / BT, tn = (N + BT - 1) / BT;
  const int splits = choose_splits(tm * tn, K);
  const int64_t chunk = chunk_for(K, splits);
  // LDS-DMA kernel: whole slices only, at least the 3 slices its prologue puts in flight
  const bool dma = dma_enabled() && (chunk % BK) == 0 && chunk >= 3 * BK && (int64_t)splits * chunk == K;
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
  if (dma)
    hipLaunchKernelGGL(gemm_fwd_dma_kernel, grid, dim3(kBlock), 0, s, F, (int64_t)K, W, (int64_t)K, M, (int64_t)N, chunk,
                       out, (int64_t)N, sstride, kb, kk, scale);
  else
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

extern "C" size_t mmssl_linear_split_workspace_bytes(int64_t M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 16;
  const int splits = choose_splits(((M + BT - 1) / BT) * ((N + BT - 1) / BT), K);
  return splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 16;
}

extern "C" int mmssl_split_bf16_f32(const float* X, int64_t n, uint16_t* hi, uint16_t* lo, void* stream) {
  if (n < 0 || (n & 3) || (n > 0 && (!X || !hi || !lo))) return MMSSL_E_BADARG;
  if (((uintptr_t)X & 15) || (((uintptr_t)hi | (uintptr_t)lo) & 7)) return MMSSL_E_BADARG;
  if (n == 0) return 0;
  int64_t nb = (n / 4 + kBlock - 1) / kBlock;
  nb = nb > 4096 ? 4096 : nb;
  hipLaunchKernelGGL(split_pair_kernel, dim3((unsigned)nb), dim3(kBlock), 0, as_stream(stream),
                     reinterpret_cast<const float4*>(X), n / 4, reinterpret_cast<ushort4*>(hi),
                     reinterpret_cast<ushort4*>(lo));
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t mmssl_split_transpose_workspace_bytes(int64_t M, int N) {
  if (M <= 0 || N <= 0) return 16;
  return (size_t)((M + 63) / 64) * (size_t)N * sizeof(float) + 16;
}

extern "C" int mmssl_split_transpose_bf16_f32(const float* G, const uint8_t* keep, float scale, int64_t M, int N,
                                              int64_t Mp, uint16_t* T_hi, uint16_t* T_lo, float* colsum,
                                              void* workspace, size_t workspace_bytes, void* stream) {
  if (M <= 0 || N <= 0 || !G || !T_hi || !T_lo || Mp < M || (Mp & 63)) return MMSSL_E_BADARG;
  if ((N & 3) || N > 256) return MMSSL_E_UNSUPP;
  if (!workspace || workspace_bytes < mmssl_split_transpose_workspace_bytes(Mp, N)) return MMSSL_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  float* colpart = reinterpret_cast<float*>(workspace);
  const unsigned nbm = (unsigned)(Mp / 64);
  hipLaunchKernelGGL(split_transpose_kernel, dim3(nbm, (unsigned)((N + 63) / 64)), dim3(kBlock), 0, s, G, keep, scale,
                     M, N, Mp, T_hi, T_lo, colpart);
  MMSSL_LAUNCH_CHECK();
  if (colsum) {
    hipLaunchKernelGGL(colsum_stage2, dim3(1), dim3(kBlock), 0, s, colpart, (int)nbm, N, colsum);
    MMSSL_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int mmssl_linear_split_f32(const uint16_t* A_hi, const uint16_t* A_lo, const uint16_t* B_hi,
                                      const uint16_t* B_lo, const float* b, const uint8_t* keep, float scale,
                                      int64_t M, int K, int N, float* Y, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  if (M <= 0 || K <= 0 || N <= 0 || !A_hi || !A_lo || !B_hi || !B_lo || !Y) return MMSSL_E_BADARG;
  if ((K % BK) || (N & 3)) return MMSSL_E_UNSUPP;
  if ((b || keep) && N > 256) return MMSSL_E_UNSUPP;
  if (((uintptr_t)A_hi | (uintptr_t)A_lo | (uintptr_t)B_hi | (uintptr_t)B_lo | (uintptr_t)Y) & 15) return MMSSL_E_BADARG;
  hipStream_t s = as_stream(stream);
  const int64_t tm = (M + BT - 1) / BT, tn = (N + BT - 1) / BT;
  int splits = choose_splits(tm * tn, K);
  int64_t chunk = chunk_for(K, splits);
  while (splits > 1 && ((int64_t)splits * chunk != K || chunk < 3 * BK)) {     // whole slices, >= 3 per split
    --splits;
    chunk = chunk_for(K, splits);
  }
  if ((int64_t)splits * chunk != K || chunk < 3 * BK) return MMSSL_E_UNSUPP;
  float* out = Y;
  if (splits > 1) {
    const size_t need = (size_t)splits * (size_t)M * (size_t)N * sizeof(float);
    if (!workspace || workspace_bytes < need) return MMSSL_E_WORKSPACE;
    out = reinterpret_cast<float*>(workspace);
  }
  hipLaunchKernelGGL(gemm_split_kernel, dim3((unsigned)tm, (unsigned)tn, (unsigned)splits), dim3(kBlock), 0, s, A_hi,
                     A_lo, (int64_t)K, B_hi, B_lo, (int64_t)K, M, (int64_t)N, chunk, out, (int64_t)N,
                     splits > 1 ? (int64_t)M * N : (int64_t)0, splits > 1 ? (const float*)nullptr : b,
                     splits > 1 ? (const uint8_t*)nullptr : keep, scale);
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
  static const int v = getenv("MMSSL_WGRAD_V") ? atoi(getenv("MMSSL_WGRAD_V")) : 10;
  return v == 10 && K % 64 == 0 && N % 64 == 0 && M * (int64_t)(K > N ? K : N) < ((int64_t)1 << 30) &&
         K < (1 << 18) && N < (1 << 18);
}
inline WgPlan wg10_plan(int64_t M, int K, int N) {
  static const int target = getenv("MMSSL_WG10_BLOCKS") ? atoi(getenv("MMSSL_WG10_BLOCKS")) : 512;
  WgPlan p;
  p.tk = K / 64;
  p.tn = N / 64;
  const int64_t tiles = (int64_t)p.tk * p.tn;
  static const int max_sp = getenv("MMSSL_WG10_MAXSP") ? atoi(getenv("MMSSL_WG10_MAXSP")) : 8;
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

namespace {
// row ranges of the transposed-feature forward: the count that minimises rounds x (work per block + epilogue) on two
// block slots per CU, ranges of at least 128 reduction rows
inline int ft_splits(int64_t tiles, int64_t K) {
  int best = 1;
  double best_cost = 1e30;
  for (int sp = 1; sp <= 16; ++sp) {
    const int64_t rows = ((K + sp - 1) / sp + 15) / 16 * 16;
    if (sp > 1 && rows < 128) break;
    const double rounds = (double)((tiles * sp + 511) / 512);
    const double cost = rounds * ((double)rows / 16.0 * 1024.0 + 4000.0) + (double)sp * 1500.0;
    if (cost < best_cost) {
      best_cost = cost;
      best = sp;
    }
  }
  if (const char* e = getenv("MMSSL_FT_SPLITS")) {
    const int f = atoi(e);
    if (f >= 1 && f <= 64) best = f;
  }
  return best;
}
inline bool ft_usable(int64_t M, int K, int N, int64_t Mp) {
  return N % 64 == 0 && K % 4 == 0 && Mp % 64 == 0 && Mp >= M && (int64_t)K * Mp < ((int64_t)1 << 30) && Mp < (1 << 18) &&
         N < (1 << 18);
}
}  // namespace

extern "C" size_t mmssl_linear_ft_workspace_bytes(int64_t M, int K, int N, int64_t Mp) {
  if (M <= 0 || K <= 0 || N <= 0 || !ft_usable(M, K, N, Mp)) return 0;
  const int sp = ft_splits((Mp / 64) * (N / 64), K);
  return ((size_t)K * N + (size_t)sp * N * Mp) * sizeof(float) + 16;
}

extern "C" int mmssl_linear_ft_f32(const float* FT, int64_t Mp, const float* W, const float* b, const uint8_t* keep,
                                   float scale, int64_t M, int K, int N, float* Y, void* workspace,
                                   size_t workspace_bytes, void* stream) {
  if (M <= 0 || K <= 0 || N <= 0 || !FT || !W || !Y) return MMSSL_E_BADARG;
  if (!ft_usable(M, K, N, Mp)) return MMSSL_E_UNSUPP;
  if (((uintptr_t)FT | (uintptr_t)W | (uintptr_t)Y | (uintptr_t)keep) & 15) return MMSSL_E_BADARG;
  const size_t need = mmssl_linear_ft_workspace_bytes(M, K, N, Mp);
  if (!workspace || workspace_bytes < need) return MMSSL_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  float* WT = reinterpret_cast<float*>(workspace);                 // [K][N]
  float* P = WT + (size_t)K * N;                                    // [sp][N][Mp]
  hipLaunchKernelGGL(w_transpose_kernel, dim3((unsigned)((K + 63) / 64), (unsigned)(N / 64)), dim3(kBlock), 0, s, W, N, K,
                     WT);
  MMSSL_LAUNCH_CHECK();
  const int tk = (int)(Mp / 64), tn = N / 64;
  const int sp0 = ft_splits((int64_t)tk * tn, K);
  const int64_t ms = (((int64_t)K + sp0 - 1) / sp0 + 15) / 16 * 16;
  const int sp = (int)((K + ms - 1) / ms);
  // the weight-gradient kernel with (gY, F, M, K) := (WT, FT, K, Mp): out[n][m] = sum_k WT[k][n] FT[k][m]
  auto* kern = gemm_nt() ? wgrad10_kernel<false, true, 8> : wgrad10_kernel<false, false, 8>;
  hipLaunchKernelGGL(kern, dim3((unsigned)tk, (unsigned)tn, (unsigned)sp), dim3(kBlock), 0, s, WT, (const uint8_t*)nullptr,
                     1.f, FT, (int64_t)K, (int)Mp, N, ms, P, (int64_t)N * Mp, (float*)nullptr);
  MMSSL_LAUNCH_CHECK();
  hipLaunchKernelGGL(ft_reduce_kernel, dim3((unsigned)tk, (unsigned)tn), dim3(kBlock), 0, s, P, sp, (int64_t)N * Mp, Mp, M,
                     N, Y, b, keep, scale);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_linear_wgrad_parts_f32(const float* gY, const uint8_t* keep, float scale, const float* F, int64_t M,
                                            int K, int N, void* workspace, size_t workspace_bytes, int* n_parts,
                                            int64_t* bias_offset, void* stream) {
  if (M <= 0 || K <= 0 || N <= 0 || !gY || !F || !n_parts || !bias_offset) return MMSSL_E_BADARG;
  if (!wg10_usable(M, K, N)) return MMSSL_E_UNSUPP;
  if (!workspace || workspace_bytes < mmssl_linear_wgrad_workspace_bytes(M, K, N)) return MMSSL_E_WORKSPACE;
  if (((uintptr_t)gY | (uintptr_t)F | (uintptr_t)workspace) & 15) return MMSSL_E_BADARG;
  if ((uintptr_t)keep & 3) return MMSSL_E_BADARG;
  const WgPlan p = wg10_plan(M, K, N);
  float* P = reinterpret_cast<float*>(workspace);
  float* bpart = P + (size_t)p.sp * N * K;
  auto* kern = keep ? (gemm_nt() ? wgrad10_kernel<true, true, 8> : wgrad10_kernel<true, false, 8>)
                    : (gemm_nt() ? wgrad10_kernel<false, true, 8> : wgrad10_kernel<false, false, 8>);
  hipLaunchKernelGGL(kern, dim3((unsigned)p.tk, (unsigned)p.tn, (unsigned)p.sp), dim3(kBlock), 0, as_stream(stream), gY,
                     keep, scale, F, M, K, N, p.ms, P, (int64_t)N * K, bpart);
  MMSSL_LAUNCH_CHECK();
  *n_parts = p.sp;
  *bias_offset = (int64_t)p.sp * N * K;
  return 0;
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
    auto* kern = keep ? (gemm_nt() ? wgrad10_kernel<true, true, 8> : wgrad10_kernel<true, false, 8>)
                      : (gemm_nt() ? wgrad10_kernel<false, true, 8> : wgrad10_kernel<false, false, 8>);
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
