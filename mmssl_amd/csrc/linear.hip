// Modality projection GEMM on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32:
// exact fp32, 157 TF peak = the only MFMA use on this path; everything else is HBM-bound).
// Replaces nn.Linear image_trans / text_trans + nn.Dropout and their autograd
// (/root/reference/MMSSL/Models.py:28-29, 54, 173-174).
//
//   forward : Y[M,N]  = dropout(F[M,K] . W[N,K]^T + b)          (F streamed once from HBM)
//   wgrad   : gW[N,K] = gY[M,N]^T . F[M,K],  gb[N] = colsum(gY)
//
// One kernel body: a 256-thread block (2x2 waves, one 32x32 MFMA accumulator each) owns a
// 64x64 output tile and walks its reduction range in 32-deep slices. Operand slices are
// fetched global->registers one slice ahead (full 128-B row segments), written to a
// double-buffered k-major LDS image (row stride 65 floats: conflict-free ds_write_b32 of the
// transposed slice and conflict-free ds_read_b32 of the MFMA fragments), one barrier per
// slice. The reduction dimension is split over blockIdx.z so that >= ~4 blocks per CU exist
// even for M = 18K; split partials are summed in a fixed order by a small epilogue kernel
// that also applies bias + dropout (deterministic, no float atomics).
#include <cstdlib>

#include "common.hpp"

using namespace mmssl;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int BT = 64;    // output tile edge
constexpr int BK = 32;    // reduction slice
constexpr int LD = 65;    // LDS row stride (floats)

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
      S[k * LD + i + 0] = r[p].x;
      S[k * LD + i + 1] = r[p].y;
      S[k * LD + i + 2] = r[p].z;
      S[k * LD + i + 3] = r[p].w;
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
  __shared__ float As[2][BK * LD];
  __shared__ float Bs[2][BK * LD];
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
// v2 kernel: each wave owns a 64x64 output patch (2x2 MFMA tiles, four INDEPENDENT 32x32 accumulators),
// so one A and one B fragment feed two MFMAs each (MFMA : ds_read = 1 : 1 instead of 1 : 2) and a
// 32-deep slice carries 64 MFMAs (4096 pipe cycles) per wave between barriers instead of 16.
// The 4 waves of a block are stacked along the LONG output dimension: 256x64 (forward: rows of F)
// or 64x256 (wgrad: columns of gW). One LDS image per operand (~41 KB per block -> 3 blocks per CU),
// next slice prefetched global->registers during the MFMA phase, two barriers per slice.
// ======================================================================================
template <bool DIRECT, int BT>
__device__ __forceinline__ void fetch_tile(const float* __restrict__ P, int64_t ld, int64_t i0, int64_t I,
                                           int64_t kk0, int64_t kk_end, float4 (&r)[BT / 32]) {
  constexpr int NF = BT / 32;        // float4 per thread: BT * BK / (256 * 4)
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < NF; ++p) {
    r[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!DIRECT) {
      const int64_t i = i0 + (tid >> 3) + 32 * p;
      const int64_t kk = kk0 + 4 * (tid & 7);
      if (i < I && kk < kk_end) r[p] = *reinterpret_cast<const float4*>(P + i * ld + kk);
    } else {
      constexpr int LPRW = BT / 4;            // lanes per slice row
      const int64_t kk = kk0 + tid / LPRW + (kBlock / LPRW) * p;
      const int64_t i = i0 + 4 * (tid % LPRW);
      if (kk < kk_end && i < I) r[p] = *reinterpret_cast<const float4*>(P + kk * ld + i);
    }
  }
}

template <bool DIRECT, int BT>
__device__ __forceinline__ void store_tile(float* __restrict__ S, const float4 (&r)[BT / 32]) {
  constexpr int NF = BT / 32;
  constexpr int LD = BT + (DIRECT ? 4 : 1);
  const int tid = threadIdx.x;
#pragma unroll
  for (int p = 0; p < NF; ++p) {
    if (!DIRECT) {
      const int i = (tid >> 3) + 32 * p, k = 4 * (tid & 7);
      S[(k + 0) * LD + i] = r[p].x;
      S[(k + 1) * LD + i] = r[p].y;
      S[(k + 2) * LD + i] = r[p].z;
      S[(k + 3) * LD + i] = r[p].w;
    } else {
      constexpr int LPRW = BT / 4;
      const int k = tid / LPRW + (kBlock / LPRW) * p, i = 4 * (tid % LPRW);
      *reinterpret_cast<float4*>(S + k * LD + i) = r[p];
    }
  }
}

template <bool DIRECT, int WM, int WN>
__global__ __launch_bounds__(kBlock) void gemm_w64_kernel(const float* __restrict__ A, int64_t lda,
                                                          const float* __restrict__ B, int64_t ldb, int64_t I,
                                                          int64_t J, int64_t KK, int64_t kk_chunk,
                                                          float* __restrict__ C, int64_t ldc, int64_t split_stride,
                                                          const float* __restrict__ bias,
                                                          const uint8_t* __restrict__ keep, float scale) {
  static_assert(WM * WN == 4, "four waves per block");
  constexpr int BTI = WM * 64, BTJ = WN * 64;
  constexpr int LDI = BTI + (DIRECT ? 4 : 1), LDJ = BTJ + (DIRECT ? 4 : 1);
  __shared__ __attribute__((aligned(16))) float As[BK * LDI];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDJ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int64_t i0 = (int64_t)blockIdx.x * BTI, j0 = (int64_t)blockIdx.y * BTJ;
  const int64_t kk_beg = (int64_t)blockIdx.z * kk_chunk;
  const int64_t kk_end = min(KK, kk_beg + kk_chunk);
  const int nk = (int)((kk_end - kk_beg + BK - 1) / BK);
  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float4 ra[BTI / 32], rb[BTJ / 32];
  if (nk > 0) {
    fetch_tile<DIRECT, BTI>(A, lda, i0, I, kk_beg, kk_end, ra);
    fetch_tile<DIRECT, BTJ>(B, ldb, j0, J, kk_beg, kk_end, rb);
    store_tile<DIRECT, BTI>(As, ra);
    store_tile<DIRECT, BTJ>(Bs, rb);
  }
  __syncthreads();
  const float* as = As + (lane >> 5) * LDI + wm * 64 + (lane & 31);
  const float* bs = Bs + (lane >> 5) * LDJ + wn * 64 + (lane & 31);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {   // next slice: global -> registers while this slice is multiplied
      fetch_tile<DIRECT, BTI>(A, lda, i0, I, kk_beg + (int64_t)(kt + 1) * BK, kk_end, ra);
      fetch_tile<DIRECT, BTJ>(B, ldb, j0, J, kk_beg + (int64_t)(kt + 1) * BK, kk_end, rb);
    }
#pragma unroll
    for (int s = 0; s < BK / 2; ++s) {
      const float a0 = as[2 * s * LDI], a1 = as[2 * s * LDI + 32];
      const float b0 = bs[2 * s * LDJ], b1 = bs[2 * s * LDJ + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();                       // every wave is done reading this slice
    if (kt + 1 < nk) {
      store_tile<DIRECT, BTI>(As, ra);
      store_tile<DIRECT, BTJ>(Bs, rb);
    }
    __syncthreads();
  }
  float* Cp = C + (int64_t)blockIdx.z * split_stride;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int64_t col = j0 + wn * 64 + b * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = i0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row < I && col < J) {
          float v = acc[a][b][r];
          if (bias) v += bias[col];
          if (keep) v = keep[row * J + col] ? v * scale : 0.f;
          Cp[row * ldc + col] = v;
        }
      }
    }
}

// ======================================================================================
// v3 forward kernel: barrier-free, LDS-free. Every WAVE streams its own 32 x N output tile over a K range:
// both MFMA operands (v_mfma_f32_16x16x4_f32) are loaded straight from global memory in fragment order —
// lane (i = l&15, q = l>>4) reads the float4 F[row i][k + 4q .. +3]; element j of that float4 is the
// k-slot-q operand of MFMA step j, and W is read with the same (q, j) -> k map, so the sum over the 4
// slots and the 4 steps covers 16 consecutive k exactly once. Two register sets are ping-ponged (next
// 16 k in flight while the current 16 are multiplied); waves never synchronise, so load latency of one
// wave hides behind the MFMAs of the others. The 4 waves of a block take 4 consecutive row tiles of the
// SAME K range, so their W fragments hit in the CU's L1. Rows past M are clamped (valid reads, results
// discarded); requires K % 32 == 0 and N in {64, 128}; partials go through splitk_reduce_kernel.
// ======================================================================================
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(kBlock) void gemm_fwd_direct_kernel(const float* __restrict__ F,
                                                                 const float* __restrict__ W, int64_t M, int K,
                                                                 int k_chunk, float* __restrict__ P) {
  constexpr int N = NT * 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
  if (m0 >= M) return;                                   // no barriers in this kernel
  const int k_beg = blockIdx.y * k_chunk;
  const int k_end = min(K, k_beg + k_chunk);
  const float* ap[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    int64_t r = m0 + mt * 16 + li;
    r = r < M ? r : M - 1;
    ap[mt] = F + r * K + 4 * lq;
  }
  const float* bp = W + (int64_t)li * K + 4 * lq;        // n-tile nt adds nt*16*K
  floatx4 acc[2][NT];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mt][nt][r] = 0.f;
  float4 a0[2], b0[NT], a1[2], b1[NT];
  auto load = [&](float4 (&a)[2], float4 (&b)[NT], int k) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) a[mt] = *reinterpret_cast<const float4*>(ap[mt] + k);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = *reinterpret_cast<const float4*>(bp + (int64_t)nt * 16 * K + k);
  };
  auto mul = [&](const float4 (&a)[2], const float4 (&b)[NT]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float av = j == 0 ? a[mt].x : j == 1 ? a[mt].y : j == 2 ? a[mt].z : a[mt].w;
          const float bv = j == 0 ? b[nt].x : j == 1 ? b[nt].y : j == 2 ? b[nt].z : b[nt].w;
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[mt][nt], 0, 0, 0);
        }
  };
  load(a0, b0, k_beg);
  int k = k_beg;
  for (; k + 32 < k_end; k += 32) {      // steady state: no conditional loads, so the counted vmcnt waits
    load(a1, b1, k + 16);                //  only ever wait for the set that is about to be multiplied
    mul(a0, b0);
    load(a0, b0, k + 32);
    mul(a1, b1);
  }
  load(a1, b1, k + 16);                  // peeled last 32
  mul(a0, b0);
  mul(a1, b1);
  // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
  float* Pp = P + (int64_t)blockIdx.y * M * N;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t row = m0 + mt * 16 + lq * 4 + r;
      if (row < M) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) Pp[row * N + nt * 16 + li] = acc[mt][nt][r];
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

// 1 (default): 64x64 block tile, 32x32 per wave.  2: 64x64 per wave (gemm_w64_kernel).
// Measured on MI355X (Baby image projection, K=4096): v1 132 / 127 us (fwd / wgrad), v2 139 / 126 us:
// both sit at ~75 TF because the limiter is the bulk-synchronous load -> wait -> MFMA cadence
// (tools/overlap_probe.py), not the LDS:MFMA ratio.
inline int gemm_version() {
  static int v = getenv("MMSSL_GEMM_V") ? atoi(getenv("MMSSL_GEMM_V")) : 1;
  return v;
}
// v2: ~3 blocks of 41 KB LDS per CU
inline int choose_splits_v2(int64_t tiles, int64_t KK) {
  const int64_t slices = (KK + BK - 1) / BK;
  if (const char* e = getenv("MMSSL_GEMM_SPLITS")) {
    const int64_t f = atoi(e);
    if (f >= 1) return (int)(f > slices ? slices : f);
  }
  static const int target = getenv("MMSSL_GEMM_TARGET_BLOCKS") ? atoi(getenv("MMSSL_GEMM_TARGET_BLOCKS")) : 768;
  int64_t s = (target + tiles - 1) / tiles;
  const int64_t max_s = slices / 4 > 0 ? slices / 4 : 1;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return (int)s;
}

// v3: one wave per 32-row tile and K range; aim for ~4-5 waves per SIMD (4096+ waves), K ranges
// multiples of 32 and at least 128 deep
inline int direct_splits(int64_t M, int K) {
  if (const char* e = getenv("MMSSL_GEMM_SPLITS")) {
    const int f = atoi(e);
    if (f >= 1) return f > K / 32 ? K / 32 : f;
  }
  const int64_t tiles = (M + 31) / 32;
  int64_t s = (4608 + tiles - 1) / tiles;
  const int64_t max_s = K / 128 > 0 ? K / 128 : 1;
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  return (int)s;
}
inline int direct_chunk(int K, int splits) { return (((K / 32) + splits - 1) / splits) * 32; }

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
  const bool v2 = gemm_version() == 2;
  const int64_t tiles = v2 ? ((M + 255) / 256) * ((N + 63) / 64) : ((M + BT - 1) / BT) * ((N + BT - 1) / BT);
  int splits = v2 ? choose_splits_v2(tiles, K) : choose_splits(tiles, K);
  if (gemm_version() == 3 && (K % 32) == 0 && (N == 64 || N == 128)) {
    const int s3 = direct_splits(M, K);
    splits = s3 > splits ? s3 : splits;
    if (splits < 2) splits = 2;            // v3 always goes through the partial buffer
  }
  return splits > 1 ? (size_t)splits * (size_t)M * (size_t)N * sizeof(float) : 16;
}

extern "C" int mmssl_linear_f32(const float* F, const float* W, const float* b, const uint8_t* keep, float scale,
                                int64_t M, int K, int N, float* Y, void* workspace, size_t workspace_bytes,
                                void* stream) {
  if (M < 0 || K <= 0 || N <= 0 || (M > 0 && (!F || !W || !Y))) return MMSSL_E_BADARG;
  if ((K & 3) || (N & 3) || N > 256) return MMSSL_E_UNSUPP;
  if (M == 0) return 0;
  if (((uintptr_t)F | (uintptr_t)W | (uintptr_t)Y) & 15) return MMSSL_E_BADARG;
  hipStream_t s = as_stream(stream);
  if (gemm_version() == 3 && (K % 32) == 0 && (N == 64 || N == 128)) {
    const int sp = direct_splits(M, K);
    const int kc = direct_chunk(K, sp);
    const size_t need3 = (size_t)sp * (size_t)M * (size_t)N * sizeof(float);
    if (!workspace || workspace_bytes < need3) return MMSSL_E_WORKSPACE;
    float* P3 = reinterpret_cast<float*>(workspace);
    const dim3 grid3((unsigned)((M + 127) / 128), (unsigned)sp);
    if (N == 64)
      hipLaunchKernelGGL((gemm_fwd_direct_kernel<4>), grid3, dim3(kBlock), 0, s, F, W, M, K, kc, P3);
    else
      hipLaunchKernelGGL((gemm_fwd_direct_kernel<8>), grid3, dim3(kBlock), 0, s, F, W, M, K, kc, P3);
    MMSSL_LAUNCH_CHECK();
    const int64_t total3 = M * N;
    int64_t nb3 = (total3 / 4 + kBlock - 1) / kBlock;
    nb3 = nb3 > 4096 ? 4096 : (nb3 < 1 ? 1 : nb3);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb3), dim3(kBlock), 0, s, P3, sp, total3, (int64_t)N, b,
                       keep, scale, Y);
    MMSSL_LAUNCH_CHECK();
    return 0;
  }
  if (gemm_version() == 2) {
    const int64_t tm2 = (M + 255) / 256, tn2 = (N + 63) / 64;
    const int sp = choose_splits_v2(tm2 * tn2, K);
    const int64_t ch = chunk_for(K, sp);
    if (sp == 1) {
      hipLaunchKernelGGL((gemm_w64_kernel<false, 4, 1>), dim3((unsigned)tm2, (unsigned)tn2, 1), dim3(kBlock), 0, s, F,
                         (int64_t)K, W, (int64_t)K, M, (int64_t)N, (int64_t)K, ch, Y, (int64_t)N, (int64_t)0, b, keep,
                         scale);
      MMSSL_LAUNCH_CHECK();
      return 0;
    }
    const size_t need2 = (size_t)sp * (size_t)M * (size_t)N * sizeof(float);
    if (!workspace || workspace_bytes < need2) return MMSSL_E_WORKSPACE;
    float* P2 = reinterpret_cast<float*>(workspace);
    hipLaunchKernelGGL((gemm_w64_kernel<false, 4, 1>), dim3((unsigned)tm2, (unsigned)tn2, (unsigned)sp), dim3(kBlock),
                       0, s, F, (int64_t)K, W, (int64_t)K, M, (int64_t)N, (int64_t)K, ch, P2, (int64_t)N,
                       (int64_t)M * N, (const float*)nullptr, (const uint8_t*)nullptr, 1.f);
    MMSSL_LAUNCH_CHECK();
    const int64_t total2 = M * N;
    int64_t nb2 = (total2 / 4 + kBlock - 1) / kBlock;
    nb2 = nb2 > 4096 ? 4096 : (nb2 < 1 ? 1 : nb2);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb2), dim3(kBlock), 0, s, P2, sp, total2, (int64_t)N, b,
                       keep, scale, Y);
    MMSSL_LAUNCH_CHECK();
    return 0;
  }
  const int64_t tm = (M + BT - 1) / BT, tn = (N + BT - 1) / BT;
  const int splits = choose_splits(tm * tn, K);
  const int64_t chunk = chunk_for(K, splits);
  if (splits == 1) {
    hipLaunchKernelGGL((gemm64_kernel<false>), dim3((unsigned)tm, (unsigned)tn, 1), dim3(kBlock), 0, s, F,
                       (int64_t)K, W, (int64_t)K, M, (int64_t)N, (int64_t)K, chunk, Y, (int64_t)N, (int64_t)0, b,
                       keep, scale, (const uint8_t*)nullptr, 1.f);
    MMSSL_LAUNCH_CHECK();
    return 0;
  }
  const size_t need = (size_t)splits * (size_t)M * (size_t)N * sizeof(float);
  if (!workspace || workspace_bytes < need) return MMSSL_E_WORKSPACE;
  float* P = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL((gemm64_kernel<false>), dim3((unsigned)tm, (unsigned)tn, (unsigned)splits), dim3(kBlock), 0,
                     s, F, (int64_t)K, W, (int64_t)K, M, (int64_t)N, (int64_t)K, chunk, P, (int64_t)N,
                     (int64_t)M * N, (const float*)nullptr, (const uint8_t*)nullptr, 1.f, (const uint8_t*)nullptr, 1.f);
  MMSSL_LAUNCH_CHECK();
  const int64_t total = M * N;
  int64_t nb = (total / 4 + kBlock - 1) / kBlock;
  nb = nb > 4096 ? 4096 : (nb < 1 ? 1 : nb);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(kBlock), 0, s, P, splits, total, (int64_t)N,
                     b, keep, scale, Y);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t mmssl_linear_wgrad_workspace_bytes(int64_t M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 16;
  // large enough for either kernel version (the masked-fetch form always uses v1)
  const int s1 = choose_splits(((N + BT - 1) / BT) * ((K + BT - 1) / BT), M);
  const int s2 = choose_splits_v2(((N + 63) / 64) * ((K + 255) / 256), M);
  const int splits = s1 > s2 ? s1 : s2;
  const size_t part = splits > 1 ? (size_t)splits * (size_t)N * (size_t)K * sizeof(float) : 0;
  return part + (size_t)kColsumBlocks * (size_t)N * sizeof(float) + 16;
}

extern "C" int mmssl_linear_wgrad_f32(const float* gY, const uint8_t* keep, float scale, const float* F, int64_t M,
                                      int K, int N, float* gW, float* gb, void* workspace,
                                      size_t workspace_bytes, void* stream) {
  if (M <= 0 || K <= 0 || N <= 0 || !gY || !F || !gW) return MMSSL_E_BADARG;
  if ((K & 3) || (N & 3) || N > 256) return MMSSL_E_UNSUPP;
  if (!workspace || workspace_bytes < mmssl_linear_wgrad_workspace_bytes(M, K, N)) return MMSSL_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  const bool v2 = gemm_version() == 2 && keep == nullptr;
  const int64_t tn = v2 ? (N + 63) / 64 : (N + BT - 1) / BT, tk = v2 ? (K + 255) / 256 : (K + BT - 1) / BT;
  const int splits = v2 ? choose_splits_v2(tn * tk, M) : choose_splits(tn * tk, M);
  const int64_t chunk = chunk_for(M, splits);
  float* ws = reinterpret_cast<float*>(workspace);
  float* colpart = ws;                                   // [kColsumBlocks][N]
  float* P = ws + (size_t)kColsumBlocks * N;             // [splits][N][K]
  if (v2) {
    float* dst = splits == 1 ? gW : P;
    hipLaunchKernelGGL((gemm_w64_kernel<true, 1, 4>), dim3((unsigned)tn, (unsigned)tk, (unsigned)splits), dim3(kBlock),
                       0, s, gY, (int64_t)N, F, (int64_t)K, (int64_t)N, (int64_t)K, M, chunk, dst, (int64_t)K,
                       splits == 1 ? (int64_t)0 : (int64_t)N * K, (const float*)nullptr, (const uint8_t*)nullptr, 1.f);
    MMSSL_LAUNCH_CHECK();
    if (splits > 1) {
      const int64_t total = (int64_t)N * K;
      int64_t nb = (total / 4 + kBlock - 1) / kBlock;
      nb = nb > 4096 ? 4096 : (nb < 1 ? 1 : nb);
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)nb), dim3(kBlock), 0, s, P, splits, total, (int64_t)K,
                         (const float*)nullptr, (const uint8_t*)nullptr, 1.f, gW);
      MMSSL_LAUNCH_CHECK();
    }
  } else
  // gW[n][k] = sum_m gY[m][n] * F[m][k]: A = gY as [kk=m][i=n], B = F as [kk=m][j=k]
  if (splits == 1) {
    hipLaunchKernelGGL((gemm64_kernel<true>), dim3((unsigned)tn, (unsigned)tk, 1), dim3(kBlock), 0, s, gY,
                       (int64_t)N, F, (int64_t)K, (int64_t)N, (int64_t)K, M, chunk, gW, (int64_t)K, (int64_t)0,
                       (const float*)nullptr, (const uint8_t*)nullptr, 1.f, keep, scale);
    MMSSL_LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL((gemm64_kernel<true>), dim3((unsigned)tn, (unsigned)tk, (unsigned)splits), dim3(kBlock),
                       0, s, gY, (int64_t)N, F, (int64_t)K, (int64_t)N, (int64_t)K, M, chunk, P, (int64_t)K,
                       (int64_t)N * K, (const float*)nullptr, (const uint8_t*)nullptr, 1.f, keep, scale);
    MMSSL_LAUNCH_CHECK();
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
