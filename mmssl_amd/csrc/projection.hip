// Grouped modality projection for gfx950 (MI355X): ALL modality problems of one step in ONE stream-K launch,
// forward and weight gradient (fp32 MFMA, exact fp32).
//
// Replaces nn.Linear image_trans / text_trans (+ nn.Dropout) and their autograd for the whole modality list at once
// (/root/reference/MMSSL/Models.py:28-29, 54, 173-174):
//
//   forward  Y[:, 64g:64g+64] = dropout(F_g[M,K_g] . W_g[64,K_g]^T + b_g)        g = 0 .. n_prob-1
//   wgrad    gW_g[64,K_g] = G[:, 64g:64g+64]^T . F_g[M,K_g],  gb_g = column sums of G[:, 64g:64g+64]
//            (G = the already dropout-masked output gradient: the producing SpMM applies the mask in its epilogue)
//
// Why these kernels (profiles/r03_gemm_pmc_default.txt, r03_proj_ablation.txt): the 64x64-tile kernels of
// csrc/linear.hip re-read the SMALL operand (W, 1 MB; gY, 4.7 MB) once per 64 rows / 64 columns of the big one —
// 287 MB of L2->LDS traffic next to the 301 MB of F itself — and run the two modality problems as two launches that
// stretch each other. Here a block owns a 256 x 64 output tile (8 waves), so the small operand is re-read 4x less,
// and every problem shares one launch: the (tile, 32-deep slice) units of all problems lie on one axis that is cut
// into as many equal ranges as there are CUs — one balanced tail, no second GEMM next to the first.
//
// Pipeline (one 512-thread block per CU, two waves per SIMD): a 3-stage LDS ring of 40 KB slices (256 x 32 of the long
// operand, 64 x 32 of the short one) filled by LDS-DMA three slices ahead, one raw s_barrier per slice, and the step's
// DMA issue and fragment reads slotted BETWEEN the MFMAs (one scheduling region per quarter of a slice): issued up
// front they would leave the matrix pipe idle for their whole issue time, because the two waves of a SIMD pass the
// barrier together.
//   forward  row-major images F[i][k], W[j][k]: XOR-swizzled 128-B rows, ds_read_b128 fragments, v_mfma_f32_32x32x2,
//            wave w = rows 32w..32w+31 x 64 columns (two accumulators); a wave DMAs exactly the 32 rows it reads. ONE
//            fragment register set refilled in place (135 VGPRs): the GCN chain's SpMMs that run beside this kernel
//            keep waves resident (a second set = 211 VGPRs = the whole register file at two waves per SIMD).
//   wgrad    k-major images F[m][i], G[m][j] (the reduction index m is the ROW of both operands): one DMA piece = one
//            1 KB row; v_mfma_f32_16x16x4 with the output-permutation trick — a lane's ds_read_b128 of F[m][4i'..4i'+3]
//            feeds FOUR MFMAs whose row slot i' stands for rows 4i'+e (un-permuted for free in the epilogue), so a
//            k-group of 8 MFMAs costs two LDS instructions (32x32x2 on these images needs one ds_read_b32 per operand
//            per MFMA: 25 us of the kernel, measured). Waves 4 (rows) x 2 (columns), 64 x 32 each.
// A segment switch inside a block's range drains and refills the ring (measured: keeping the DMA running across the
// switch - one pipeline over the whole range - costs more in registers and branches than the ~2 us per switch it saves).
// (Also measured and dropped: the last-arriving block adding a tile's slots itself - write-through slot stores + arrival
// tickets, no second launch. Every range of these shapes ends inside a tile, so all 26 MB of slots go out as 8-byte
// write-through stores: forward 159 instead of 114 + 21 us, weight gradient 183 instead of 140 + 18 us in the step,
// profiles/r03/step_timeline_inkernel_fixup_rejected.txt.)
// Every range leaves its accumulator image in a partial slot; the reduce kernels add a tile's slots in block order
// (deterministic) and apply the epilogue: bias + dropout (the mask either given or drawn HERE with the generator of
// mmssl_dropout_mask_u8 — no separate mask launch in front of the GEMM) / the transposed store + bias-gradient sums.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "lds_dma.hpp"

using namespace mmssl;

// Decomposition builds for tools/proj_ablate.sh (never set in the product build): bit 0 = no DMA in the steady loop,
// bit 1 = no MFMAs, bit 2 = no fragment reads.
#ifndef MMSSL_PROJ_DBG
#define MMSSL_PROJ_DBG 0
#endif
// the streamed feature operand is loaded with the non-temporal hint (read once per launch: it should not push the SpMM
// chains' tables out of L2); 0 = plain loads, for A/B builds (tools/proj_variants.sh: 117 vs 120 us forward, 133.5 vs 134.6
// weight gradient, DMA stream alone 74 vs 82 us)
#ifndef MMSSL_PROJ_NT
#define MMSSL_PROJ_NT 1
#endif
// quarters of a slice's MFMAs issued BEFORE the step's wait + barrier (their operands sit in registers since the previous
// step), so that the matrix pipe has work while the wave waits. Build option only: measured no different (123.0 / 137.0 us
// with one quarter, 124 / 138 with two, against 122.8 / 135.3: the pipe is not idle at the barrier) - the ~20 us between
// the kernel and its MFMA time are the segment switches (~3.5 us each, 2.5 per block) and launch ramp / tail.
#ifndef MMSSL_PROJ_PRESYNC
#define MMSSL_PROJ_PRESYNC 0
#endif
constexpr int kPre = MMSSL_PROJ_PRESYNC;

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int PT = 256;                          // tile rows (index of the long operand)
constexpr int PJ = 64;                           // tile columns = N (index of the short operand)
constexpr int PBK = 32;                          // reduction slice
#ifndef MMSSL_PROJ_PST
#define MMSSL_PROJ_PST 3
#endif
constexpr int PST = MMSSL_PROJ_PST;              // LDS stages (4 = the whole LDS of a CU: measured no faster, 122 / 141 us)
constexpr int kThreads = 512;                    // 8 waves
constexpr int kStageFloats = (PT + PJ) * PBK;    // 10240 floats = 40 KB
constexpr int kLdsBytes = PST * kStageFloats * 4;
constexpr int kSlotFloats = PT * PJ;             // one partial slot: the block's accumulator image (64 KB)
constexpr int kPieces = 5;                       // DMA instructions per wave per slice (4 long + 1 short)
// forward tile height: 256 rows = one 8-wave block per CU (default); 128 = two independent 4-wave blocks per CU (one
// block's barrier waits under the other's MFMAs; twice the W re-reads). Measured (tools/proj_variants.sh,
// profiles/r03/proj_variants.txt): 125.0 vs 121.5 us alone, 0.5146 vs 0.5127 ms per step - the barrier coupling of the
// eight waves is NOT what keeps the MFMA and DMA streams from overlapping fully; kept as a build option only.
#ifndef MMSSL_PROJ_FPT
#define MMSSL_PROJ_FPT 256
#endif
constexpr int FPT = MMSSL_PROJ_FPT;
constexpr int kFWaves = FPT / 32;                // a wave = 32 rows x 64 columns
constexpr int kFThreads = 64 * kFWaves;
constexpr int kFBPieces = 8 / kFWaves;           // short-operand pieces per wave: 64 rows / (8 rows per piece) / waves
constexpr int kFPieces = 4 + kFBPieces;
constexpr int kFStageFloats = (FPT + PJ) * PBK;
constexpr int kFLdsBytes = PST * kFStageFloats * 4;
constexpr int kFSlotFloats = FPT * PJ;
constexpr int kFBlocksPerCU = 256 / FPT;
static_assert(FPT == 256 || FPT == 128, "forward tile height");
constexpr int kMaxProb = MMSSL_PROJ_MAX_PROBLEMS;

struct Group {
  const float* A[kMaxProb];       // long operand (F_g)
  const float* B[kMaxProb];       // short operand (forward: W_g; wgrad: G + 64 g)
  int64_t lda[kMaxProb], ldb[kMaxProb];
  int64_t I[kMaxProb];            // extent of the tile axis (forward: M; wgrad: K_g)
  int64_t R[kMaxProb];            // reduction length (forward: K_g; wgrad: M)
  int64_t unit0[kMaxProb + 1];    // first (tile, slice) unit of problem g
  int S[kMaxProb];                // slices per tile = ceil(R / 32)
  int tile0[kMaxProb + 1];        // first tile of problem g
  int n;
};

__device__ __forceinline__ int prob_of_unit(const Group& P, int64_t u) {
  int g = 0;
  while (g + 1 < P.n && u >= P.unit0[g + 1]) ++g;
  return g;
}
__device__ __forceinline__ int prob_of_tile(const Group& P, int t) {
  int g = 0;
  while (g + 1 < P.n && t >= P.tile0[g + 1]) ++g;
  return g;
}
__device__ __forceinline__ int tile_of_unit(const Group& P, int64_t u) {
  const int g = prob_of_unit(P, u);
  return P.tile0[g] + (int)((u - P.unit0[g]) / P.S[g]);
}

struct FwdPtrs {
  const float* bias[kMaxProb];
};
struct WgradPtrs {
  float* gW[kMaxProb];
  float* gb[kMaxProb];
};
// optional AdamW update of the projection weights / biases applied by the weight-gradient epilogue itself (the rule of
// csrc/optim.hip's adamw_kernel = torch.optim.AdamW, main.py:76-80): the step's optimiser launch leaves the critical path
struct AdamSlots {
  float* W[kMaxProb];
  float* mW[kMaxProb];
  float* vW[kMaxProb];
  float* b[kMaxProb];
  float* mb[kMaxProb];
  float* vb[kMaxProb];
  const float* state;       // state[0] = step counter (see adamw_kernel); NULL = no update
  float lr, beta1, beta2, eps, wd;
  float log2_beta1, log2_beta2;
  int pre_ticked;
  // split-precision path only: the bf16 planes of problem g's weights (projx_wsplit_kernel's image) - the weight-gradient
  // epilogue rewrites the planes of the weights it has just updated, so the next forward needs no split launch
  char* wplanes[kMaxProb];
};


// what a block's range [u, u_end) does next: one segment = the slices [s0, s0 + nk) of tile `tip` of problem g
struct Segment {
  int g, tip, s0, nk;
};
__device__ __forceinline__ Segment segment_at(const Group& P, int64_t u, int64_t u_end) {
  Segment sg;
  sg.g = prob_of_unit(P, u);
  const int S = P.S[sg.g];
  const int64_t rel = u - P.unit0[sg.g];
  sg.tip = (int)(rel / S);
  sg.s0 = (int)(rel - (int64_t)sg.tip * S);
  sg.nk = (int)min((int64_t)S, sg.s0 + (u_end - u)) - sg.s0;
  return sg;
}

// waits in front of a step that reads slice kt+1: its pieces landed (mine: vmcnt; everyone's: barrier); my fragment
// reads of slice kt are done, so after the barrier stage kt % 3 may be refilled
// at most `slices` whole slices (PCS DMA instructions each) of this wave still in flight
template <int PCS = kPieces>
__device__ __forceinline__ void wait_outstanding(int slices) {
  if (slices <= 0) vm_wait_n<0>();
  else if (slices == 1) vm_wait_n<PCS>();
  else if (slices == 2) vm_wait_n<2 * PCS>();
  else if (slices == 3 || 4 * PCS > 63) vm_wait_n<3 * PCS>();
  else if (slices == 4 || 5 * PCS > 63) vm_wait_n<(4 * PCS > 63 ? 63 : 4 * PCS)>();
  else if (slices == 5 || 6 * PCS > 63) vm_wait_n<(5 * PCS > 63 ? 63 : 5 * PCS)>();
  else if (slices == 6 || 7 * PCS > 63) vm_wait_n<(6 * PCS > 63 ? 63 : 6 * PCS)>();
  else vm_wait_n<(7 * PCS > 63 ? 63 : 7 * PCS)>();
}
template <int PCS = kPieces>
__device__ __forceinline__ void step_sync(int outstanding) {
  wait_outstanding<PCS>(outstanding);
  lgkm_wait0();
  bare_barrier();
}

// =====================================================================================================================
// forward: row-major images, v_mfma_f32_32x32x2
// =====================================================================================================================
struct FragF {
  float a[16], b0[16], b1[16];      // operand values of the 16 MFMA steps of one slice (this lane's k half)
};

__global__ __launch_bounds__(kFThreads) void proj_fwd_sk_kernel(Group P, int upb, int64_t total, int max_segs,
                                                               float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float ring[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, lr = lane & 31;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  const int64_t u_begin = (int64_t)blockIdx.x * upb;
  const int64_t u_end = min(total, u_begin + upb);
  int64_t u = u_begin;
  int seg = 0;
  while (u < u_end) {
    const Segment sg = segment_at(P, u, u_end);
    const int nk = sg.nk;
    const int64_t i0 = (int64_t)sg.tip * FPT;
    const int64_t I = P.I[sg.g], lda = P.lda[sg.g], ldb = P.ldb[sg.g];
    // DMA source addresses of this wave's five pieces (slice 0 of the segment): long operand rows 32w + 8j + (lane >> 3),
    // 16-byte chunk (lane & 7) of the 128-byte row slice, XOR-swizzled; short operand rows 8w + (lane >> 3)
    const float* pa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 32 * wave + 8 * j + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      pa[j] = P.A[sg.g] + min(i0 + r, I - 1) * lda + (int64_t)sg.s0 * PBK + 4 * c;
    }
    const float* pb[kFBPieces];
#pragma unroll
    for (int j = 0; j < kFBPieces; ++j) {
      const int r = 8 * kFBPieces * wave + 8 * j + (lane >> 3);
      const int c = (lane & 7) ^ ((r >> 1) & 7);
      pb[j] = P.B[sg.g] + (int64_t)r * ldb + (int64_t)sg.s0 * PBK + 4 * c;
    }
    auto issue_piece = [&](int kt, int e) {
      const unsigned st = ring_lds + (unsigned)(kt % PST) * (kFStageFloats * 4);
      if (e < 4) {
        if (MMSSL_PROJ_NT) glds16_nt(pa[e] + (int64_t)kt * PBK, st + (unsigned)((32 * wave_u + 8 * e) * PBK * 4));
        else glds16(pa[e] + (int64_t)kt * PBK, st + (unsigned)((32 * wave_u + 8 * e) * PBK * 4));
      }
      else glds16(pb[e - 4] + (int64_t)kt * PBK, st + (unsigned)(FPT * PBK * 4 + (8 * kFBPieces * wave_u + 8 * (e - 4)) * PBK * 4));
    };
    auto issue = [&](int kt) {
#pragma unroll
      for (int e = 0; e < kFPieces; ++e) issue_piece(kt, e);
    };
    // fragment values of MFMA steps 4q .. 4q+3 of slice kt (q = 0 .. 3)
    auto read_quarter = [&](int kt, int q, FragF& f) {
      const float* st = ring + (kt % PST) * kFStageFloats;
      const int sw = (lr >> 1) & 7;
      const int pos = ((2 * q + h) ^ sw) * 4;
      const float4 va = *reinterpret_cast<const float4*>(st + (32 * wave + lr) * PBK + pos);
      const float4 v0 = *reinterpret_cast<const float4*>(st + FPT * PBK + lr * PBK + pos);
      const float4 v1 = *reinterpret_cast<const float4*>(st + FPT * PBK + (32 + lr) * PBK + pos);
      f.a[4 * q] = va.x; f.a[4 * q + 1] = va.y; f.a[4 * q + 2] = va.z; f.a[4 * q + 3] = va.w;
      f.b0[4 * q] = v0.x; f.b0[4 * q + 1] = v0.y; f.b0[4 * q + 2] = v0.z; f.b0[4 * q + 3] = v0.w;
      f.b1[4 * q] = v1.x; f.b1[4 * q + 1] = v1.y; f.b1[4 * q + 2] = v1.z; f.b1[4 * q + 3] = v1.w;
    };
    floatx16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    // One pipeline step = the 32 MFMAs of slice kt with the step's other work slotted between them, one scheduling region
    // per quarter: the DMA pieces of slice kt+3, the 8 MFMAs of the quarter, then the fragment reads of slice kt+1's
    // quarter INTO THE REGISTERS THOSE MFMAs JUST READ (one fragment set, refilled in place).
    auto mfma_quarter = [&](int q, FragF& f) {
#pragma unroll
      for (int p = 4 * q; p < 4 * q + 4; ++p) {
        if (MMSSL_PROJ_DBG & 2) {            // decomposition build: keep the operands alive with two plain FMAs
          acc0[p] = fmaf(f.a[p], f.b0[p], acc0[p]);
          acc1[p] = fmaf(f.a[p], f.b1[p], acc1[p]);
        } else {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[p], f.b0[p], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[p], f.b1[p], acc1, 0, 0, 0);
        }
      }
    };
    auto issue_quarter = [&](int kt, int q) {
#pragma unroll
      for (int e = 0; e < kFPieces; ++e)
        if (e * 4 / kFPieces == q) issue_piece(kt + PST, e);
    };
    // sync_out >= 0: wait until at most that many slices of this wave's DMA are in flight, then the block barrier
    auto step = [&](int kt, FragF& f, bool more1, bool more3, int sync_out) {
#pragma unroll
      for (int q = 0; q < kPre; ++q) {       // operands in registers since the previous step: ahead of the wait
        mfma_quarter(q, f);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (sync_out >= 0) step_sync<kFPieces>(sync_out);
#pragma unroll
      for (int q = 0; q < kPre; ++q) {
        if (more3 && !(MMSSL_PROJ_DBG & 1)) issue_quarter(kt, q);
        if (more1 && !(MMSSL_PROJ_DBG & 4)) read_quarter(kt + 1, q, f);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = kPre; q < 4; ++q) {
        if (more3 && !(MMSSL_PROJ_DBG & 1)) issue_quarter(kt, q);
        mfma_quarter(q, f);
        if (more1 && !(MMSSL_PROJ_DBG & 4)) read_quarter(kt + 1, q, f);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // the previous segment's slot stores and fragment reads must be done before the ring is refilled
    vm_wait_n<0>();
    lgkm_wait0();
    bare_barrier();
#pragma unroll
    for (int j = 0; j < PST; ++j)
      if (j < nk) issue(j);
    wait_outstanding<kFPieces>(min(nk, PST) - 1);          // slice 0 has landed
    bare_barrier();
    FragF f;
#pragma unroll
    for (int q = 0; q < 4; ++q) read_quarter(0, q, f);
    int kt = 0;
    for (; kt + PST < nk; ++kt)                  // steady state: slices kt+1 .. kt+PST exist, branch-free
      step(kt, f, true, true, PST - 2);
    for (; kt < nk; ++kt)                        // drain
      step(kt, f, kt + 1 < nk, false, kt + 1 < nk ? min(nk - kt - 2, PST - 2) : -1);
    // the segment's accumulator image -> its partial slot: plane q (0..7) holds, at thread tid, the float4 of rows
    // 32w + 8(q & 3) + 4h + {0..3}, column 32(q >> 2) + (lane & 31)
    float4* Pq = reinterpret_cast<float4*>(partials + ((size_t)blockIdx.x * max_segs + seg) * kFSlotFloats);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      Pq[q * kFThreads + tid] = make_float4(acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]);
      Pq[(4 + q) * kFThreads + tid] = make_float4(acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]);
    }
    u += nk;
    ++seg;
  }
}

// =====================================================================================================================
// weight gradient: k-major images, v_mfma_f32_16x16x4 with permuted output slots
// =====================================================================================================================
struct FragW {
  float4 a[8];        // k-group p: F[m = 4p + g][64 wi + 4 i' .. + 3]   (row slots 4 i' + e of MFMA e)
  float2 b[8];        // k-group p: G[m = 4p + g][32 wj + 2 j' .. + 1]
};

// One fragment register set refilled in place, like the forward (172 instead of 226 VGPRs: SpMM waves of the GCN chain
// stay resident beside this kernel; measured 0.558 vs 0.571 ms per step against two alternating sets).
__global__ __launch_bounds__(kThreads) void proj_wgrad_sk_kernel(Group P, int upb, int64_t total, int max_segs,
                                                                 float* __restrict__ partials,
                                                                 float* __restrict__ bpart) {
  extern __shared__ __attribute__((aligned(16))) float ring[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g4 = lane >> 4, l16 = lane & 15;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int wi = wave >> 1, wj = wave & 1;
  const int wi_u = wave_u >> 1;
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  const int64_t u_begin = (int64_t)blockIdx.x * upb;
  const int64_t u_end = min(total, u_begin + upb);
  int64_t u = u_begin;
  int seg = 0;
  while (u < u_end) {
    const Segment sg = segment_at(P, u, u_end);
    const int nk = sg.nk, s0 = sg.s0;
    const int64_t i0 = (int64_t)sg.tip * PT;
    const int64_t I = P.I[sg.g], R = P.R[sg.g], lda = P.lda[sg.g], ldb = P.ldb[sg.g];
    // long operand image [m][256]: piece e of wave w = reduction row 4w + e, one whole 1 KB row (lane -> float4 column);
    // short operand image [m][64]: wave w's piece = rows 4w + (lane >> 4), 16-byte chunk (lane & 15), stored at chunk
    // position c ^ 8 on odd rows (the fragment read below takes 2 words per lane from rows m and m + 1 in one 32-lane
    // group: unswizzled they would share banks)
    const float* pa = P.A[sg.g] + min(i0 + 4 * lane, I - 4);
    const float* pb = P.B[sg.g] + 4 * ((lane & 15) ^ (8 * ((lane >> 4) & 1)));
    auto issue_piece = [&](int kt, int e) {
      const unsigned st = ring_lds + (unsigned)(kt % PST) * (kStageFloats * 4);
      const int64_t m0 = (int64_t)(s0 + kt) * PBK;
      if (e < 4) {
        const int64_t m = min(m0 + 4 * wave_u + e, R - 1);              // rows past the end: finite data x zeroed G
        if (MMSSL_PROJ_NT) glds16_nt(pa + m * lda, st + (unsigned)((4 * wave_u + e) * PT * 4));
        else glds16(pa + m * lda, st + (unsigned)((4 * wave_u + e) * PT * 4));
      } else {
        const int64_t m = min(m0 + 4 * wave_u + (lane >> 4), R - 1);    // clamped here, zeroed at fragment read
        glds16(pb + m * ldb, st + (unsigned)(PT * PBK * 4 + 4 * wave_u * PJ * 4));
      }
    };
    auto issue = [&](int kt) {
#pragma unroll
      for (int e = 0; e < kPieces; ++e) issue_piece(kt, e);
    };
    // k-groups 2q, 2q+1 of slice kt
    auto read_quarter = [&](int kt, int q, FragW& f) {
      const float* st = ring + (kt % PST) * kStageFloats;
      const int64_t m0 = (int64_t)(s0 + kt) * PBK;
      const bool ragged = m0 + PBK > R;       // the reduction's last slice (block-uniform): rows >= R contribute 0
#pragma unroll
      for (int p = 2 * q; p < 2 * q + 2; ++p) {
        const int m = 4 * p + g4;
        f.a[p] = *reinterpret_cast<const float4*>(st + m * PT + 64 * wi + 4 * l16);
        f.b[p] = *reinterpret_cast<const float2*>(st + PT * PBK + m * PJ + 4 * ((8 * wj + (l16 >> 1)) ^ (8 * (g4 & 1))) +
                                                  2 * (l16 & 1));
        if (ragged && m0 + m >= R) f.b[p] = make_float2(0.f, 0.f);
      }
    };
    floatx4 acc[4][2];
#pragma unroll
    for (int ea = 0; ea < 4; ++ea)
#pragma unroll
      for (int eb = 0; eb < 2; ++eb) acc[ea][eb] = floatx4{0.f, 0.f, 0.f, 0.f};
    float bs0 = 0.f, bs1 = 0.f;                           // this wave's share of the bias-gradient column sums
    const bool want_bias = bpart != nullptr && sg.tip == 0;
    auto work_quarter = [&](int q, FragW& cur) {
      if (want_bias) {                 // the four row-waves of a column half share the k-groups: p % 4 == wi
#pragma unroll
        for (int p = 2 * q; p < 2 * q + 2; ++p)
          if ((p & 3) == wi_u) { bs0 += cur.b[p].x; bs1 += cur.b[p].y; }
      }
#pragma unroll
      for (int p = 2 * q; p < 2 * q + 2; ++p) {
        const float av[4] = {cur.a[p].x, cur.a[p].y, cur.a[p].z, cur.a[p].w};
        const float bv[2] = {cur.b[p].x, cur.b[p].y};
#pragma unroll
        for (int ea = 0; ea < 4; ++ea)
#pragma unroll
          for (int eb = 0; eb < 2; ++eb) {
            if (MMSSL_PROJ_DBG & 2) acc[ea][eb][0] = fmaf(av[ea], bv[eb], acc[ea][eb][0]);
            else acc[ea][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ea], bv[eb], acc[ea][eb], 0, 0, 0);
          }
      }
    };
    auto issue_quarter = [&](int kt, int q) {
#pragma unroll
      for (int e = 0; e < kPieces; ++e)
        if (e * 4 / kPieces == q) issue_piece(kt + PST, e);
    };
    // sync_out >= 0: wait until at most that many slices of this wave's DMA are in flight, then the block barrier
    auto step = [&](int kt, FragW& cur, bool more1, bool more3, int sync_out) {
#pragma unroll
      for (int q = 0; q < kPre; ++q) {       // operands in registers since the previous step: ahead of the wait
        work_quarter(q, cur);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (sync_out >= 0) step_sync(sync_out);
#pragma unroll
      for (int q = 0; q < kPre; ++q) {
        if (more3 && !(MMSSL_PROJ_DBG & 1)) issue_quarter(kt, q);
        if (more1 && !(MMSSL_PROJ_DBG & 4)) read_quarter(kt + 1, q, cur);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int q = kPre; q < 4; ++q) {
        if (more3 && !(MMSSL_PROJ_DBG & 1)) issue_quarter(kt, q);
        work_quarter(q, cur);
        if (more1 && !(MMSSL_PROJ_DBG & 4)) read_quarter(kt + 1, q, cur);      // into the registers just consumed
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    vm_wait_n<0>();
    lgkm_wait0();
    bare_barrier();
#pragma unroll
    for (int j = 0; j < PST; ++j)
      if (j < nk) issue(j);
    wait_outstanding(min(nk, PST) - 1);          // slice 0 has landed
    bare_barrier();
    FragW f;
#pragma unroll
    for (int q = 0; q < 4; ++q) read_quarter(0, q, f);
    int kt = 0;
    for (; kt + PST < nk; ++kt)                  // steady state: slices kt+1 .. kt+PST exist, branch-free
      step(kt, f, true, true, PST - 2);
    for (; kt < nk; ++kt)                        // drain
      step(kt, f, kt + 1 < nk, false, kt + 1 < nk ? min(nk - kt - 2, PST - 2) : -1);
    // acc[ea][eb][r] at lane (g, j') = C[row 64 wi + 16 g + 4 r + ea][col 32 wj + 2 j' + eb]: plane 4 eb + r holds, at
    // thread tid, the float4 of the FOUR CONSECUTIVE rows ea = 0..3 (the permuted row slots fall back into place)
    const size_t slot = (size_t)blockIdx.x * max_segs + seg;
    float4* Pq = reinterpret_cast<float4*>(partials + slot * kSlotFloats);
#pragma unroll
    for (int eb = 0; eb < 2; ++eb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Pq[(4 * eb + r) * kThreads + tid] = make_float4(acc[0][eb][r], acc[1][eb][r], acc[2][eb][r], acc[3][eb][r]);
    if (want_bias) {
      // sum over the reduction rows this lane saw (g), then the eight waves' shares meet in LDS (the ring is idle:
      // everyone is past its last fragment read after the barrier)
      bs0 += __shfl_xor(bs0, 16, kWave);
      bs0 += __shfl_xor(bs0, 32, kWave);
      bs1 += __shfl_xor(bs1, 16, kWave);
      bs1 += __shfl_xor(bs1, 32, kWave);
      lgkm_wait0();
      bare_barrier();
      if (g4 == 0) {
        ring[wave * 32 + 2 * l16] = bs0;
        ring[wave * 32 + 2 * l16 + 1] = bs1;
      }
      lgkm_wait0();
      bare_barrier();
      if (tid < PJ) {                  // column tid: half wj = tid >> 5, waves 2 wi + wj
        const int half = tid >> 5, c = tid & 31;
        bpart[slot * PJ + tid] = ((ring[(0 + half) * 32 + c] + ring[(2 + half) * 32 + c]) + ring[(4 + half) * 32 + c]) +
                                 ring[(6 + half) * 32 + c];
      }
    }
    u += nk;
    ++seg;
  }
}

// =====================================================================================================================
// epilogue kernels
// =====================================================================================================================
// Philox4x32-10 exactly as mmssl_dropout_mask_u8 draws it (csrc/optim.hip): counter = (group of 4 mask bytes,
// launch counter), key = seed; byte e of the group keeps iff the top 24 bits of word e are >= p * 2^24.
__device__ __forceinline__ uint32_t philox_keep_byte(uint64_t seed, uint64_t launch, uint64_t elem, uint32_t thr) {
  const uint64_t i = elem >> 2;
  uint32_t c[4] = {(uint32_t)i, (uint32_t)(i >> 32), (uint32_t)launch, (uint32_t)(launch >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t a = (uint64_t)0xD2511F53u * c[0];
    const uint64_t b = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(b >> 32) ^ c[1] ^ k0;
    const uint32_t n2 = (uint32_t)(a >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)b;
    c[3] = (uint32_t)a;
    c[0] = n0;
    c[2] = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return (c[elem & 3] >> 8) >= thr ? 1u : 0u;
}

// the slots of tile t, plane q, in block order (fixed: the result does not depend on the schedule). The loads of up to
// eight slots are issued together (a weight-gradient tile has ~14 of them: one load latency per slot, back to back, was
// most of the epilogue launch's time); the additions keep the block order.
template <int SLOT = kSlotFloats, int THREADS = kThreads>
__device__ __forceinline__ float4 sum_slots(const Group& P, int t, int g, int tip, int q, int upb, int max_segs,
                                            const float* __restrict__ partials, int tid) {
  const int64_t U0 = P.unit0[g] + (int64_t)tip * P.S[g], U1 = U0 + P.S[g];
  const int64_t b_first = U0 / upb, b_last = (U1 - 1) / upb;
  const int seg_first = t - tile_of_unit(P, b_first * upb);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t b0 = b_first; b0 <= b_last; b0 += 8) {
    float4 p[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t b = min(b0 + k, b_last);                    // clamped: a repeated (cached) load, not added
      const int seg = b == b_first ? seg_first : 0;
      p[k] = reinterpret_cast<const float4*>(partials + ((size_t)b * max_segs + seg) * SLOT)[q * THREADS + tid];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (b0 + k <= b_last) { v.x += p[k].x; v.y += p[k].y; v.z += p[k].z; v.w += p[k].w; }
  }
  return v;
}

// forward epilogue: grid = tiles x 8 planes
__global__ __launch_bounds__(kFThreads) void proj_fwd_reduce_kernel(Group P, int upb, int max_segs,
                                                                   const float* __restrict__ partials, int64_t M,
                                                                   float* __restrict__ Y, int64_t ldy, FwdPtrs ptrs,
                                                                   const uint8_t* __restrict__ keep_in,
                                                                   uint8_t* __restrict__ keep_out,
                                                                   const uint64_t* __restrict__ rng, float p_drop,
                                                                   float scale) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = (int)blockIdx.x >> 3, q = (int)blockIdx.x & 7;
  const int g = prob_of_tile(P, t), tip = t - P.tile0[g];
  const float4 v = sum_slots<kFSlotFloats, kFThreads>(P, t, g, tip, q, upb, max_segs, partials, tid);
  const int col = 32 * (q >> 2) + (lane & 31);
  const int64_t row0 = (int64_t)tip * FPT + 32 * wave + 8 * (q & 3) + 4 * (lane >> 5);
  const float* bias = ptrs.bias[g];
  const float bv = bias ? bias[col] : 0.f;
  const bool gen = keep_out != nullptr && rng != nullptr;
  uint64_t seed = 0, launch = 0;
  uint32_t thr = 0;
  if (gen) {
    seed = rng[0];
    launch = rng[1];
    thr = (uint32_t)(p_drop * 16777216.0f);
  }
  const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t row = row0 + c;
    if (row >= M) break;
    float x = e[c] + bv;
    const int64_t mi = ((int64_t)g * M + row) * PJ + col;
    if (gen) {
      const uint32_t k = philox_keep_byte(seed, launch, (uint64_t)mi, thr);
      keep_out[mi] = (uint8_t)k;
      x = k ? x * scale : 0.f;
    } else if (keep_in) {
      x = keep_in[mi] ? x * scale : 0.f;
    }
    Y[row * ldy + (int64_t)g * PJ + col] = x;
  }
}

__device__ __forceinline__ void adamw_update(float& p, float g, float& m, float& v, float step_size, float bc2_sqrt,
                                             float decay, float beta1, float beta2, float eps) {
  p *= decay;
  m = m + (g - m) * (1.0f - beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
  v = v * beta2 + (1.0f - beta2) * g * g;
  p -= step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
}

// weight-gradient epilogue: the tile is C[i = feature column][j = channel]; gW[j][i .. i+3] is one float4
__global__ __launch_bounds__(kThreads) void proj_wgrad_reduce_kernel(Group P, int upb, int max_segs,
                                                                     const float* __restrict__ partials,
                                                                     const float* __restrict__ bpart, WgradPtrs ptrs,
                                                                     AdamSlots ad) {
  __shared__ float sh[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t = (int)blockIdx.x >> 3, q = (int)blockIdx.x & 7;
  const int g = prob_of_tile(P, t), tip = t - P.tile0[g];
  if (ad.state) {
    // bias corrections once per block. beta^step as exp2(step * log2(beta)) on the hardware exp2 (a few ulp from the
    // powf of adamw_kernel, i.e. ~1e-7 relative on the step size): a libm powf here made every block of this short
    // epilogue wait ~3 us for one thread (27 us instead of 12 for the launch)
    if (tid == 0) {
      const float step = ad.pre_ticked ? fmaxf(ad.state[0], 1.0f) : ad.state[0] + 1.0f;
      sh[0] = ad.lr / (1.0f - __builtin_amdgcn_exp2f(step * ad.log2_beta1));
      sh[1] = sqrtf(1.0f - __builtin_amdgcn_exp2f(step * ad.log2_beta2));
    }
    __syncthreads();
  }
  const float decay = 1.0f - ad.lr * ad.wd;
  const float4 v = sum_slots(P, t, g, tip, q, upb, max_segs, partials, tid);
  const int col = 32 * (wave & 1) + 2 * (lane & 15) + (q >> 2);
  const int64_t i = (int64_t)tip * PT + 64 * (wave >> 1) + 16 * (lane >> 4) + 4 * (q & 3);
  const int64_t K = P.I[g];
  if (i < K) {
    const int64_t o = (int64_t)col * K + i;
    if (ptrs.gW[g]) *reinterpret_cast<float4*>(ptrs.gW[g] + o) = v;
    if (ad.state && ad.W[g]) {
      float4 pp = *reinterpret_cast<float4*>(ad.W[g] + o);
      float4 mm = *reinterpret_cast<float4*>(ad.mW[g] + o);
      float4 vv = *reinterpret_cast<float4*>(ad.vW[g] + o);
      adamw_update(pp.x, v.x, mm.x, vv.x, sh[0], sh[1], decay, ad.beta1, ad.beta2, ad.eps);
      adamw_update(pp.y, v.y, mm.y, vv.y, sh[0], sh[1], decay, ad.beta1, ad.beta2, ad.eps);
      adamw_update(pp.z, v.z, mm.z, vv.z, sh[0], sh[1], decay, ad.beta1, ad.beta2, ad.eps);
      adamw_update(pp.w, v.w, mm.w, vv.w, sh[0], sh[1], decay, ad.beta1, ad.beta2, ad.eps);
      *reinterpret_cast<float4*>(ad.W[g] + o) = pp;
      *reinterpret_cast<float4*>(ad.mW[g] + o) = mm;
      *reinterpret_cast<float4*>(ad.vW[g] + o) = vv;
    }
  }
  if (tip == 0 && q == 0 && tid < PJ && (ptrs.gb[g] || (ad.state && ad.b[g]))) {
    const int64_t U0 = P.unit0[g], U1 = U0 + P.S[g];
    const int64_t b_first = U0 / upb, b_last = (U1 - 1) / upb;
    float s = 0.f;
    for (int64_t b = b_first; b <= b_last; ++b) {
      const int seg = b == b_first ? t - tile_of_unit(P, b * upb) : 0;
      s += bpart[((size_t)b * max_segs + seg) * PJ + tid];
    }
    if (ptrs.gb[g]) ptrs.gb[g][tid] = s;
    if (ad.state && ad.b[g]) {
      float pp = ad.b[g][tid], mm = ad.mb[g][tid], vv = ad.vb[g][tid];
      adamw_update(pp, s, mm, vv, sh[0], sh[1], decay, ad.beta1, ad.beta2, ad.eps);
      ad.b[g][tid] = pp;
      ad.mb[g][tid] = mm;
      ad.vb[g][tid] = vv;
    }
  }
}

// =====================================================================================================================
// SPLIT-PRECISION grouped projection ("projx"): the same stream-K launch structure on the bf16 matrix pipe
// =====================================================================================================================
// fp32 MFMA runs at the fp32 VECTOR rate (157 TFLOP/s, 1/16 of the bf16 pipe): at 12 GFLOP per direction the kernels above
// are matrix-pipe bound (88 us of MFMA issue alone) with a 387 MB feature stream (62-80 us) that only partly hides under it.
// Here every fp32 operand value is cut EXACTLY into three bf16 pieces (a = hi + mid + lo: three truncations of 8 significant
// bits each cover fp32's 24) and a product a.b is evaluated as the six partial products of weight >= 2^-16
//     hi.hi + (hi.mid + mid.hi) + (hi.lo + lo.hi + mid.mid)
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: each partial product is exact in fp32 (8 x 8 significant bits), the
// three dropped ones (mid.lo, lo.mid, lo.lo) are <= 2^-23 of |a.b| - the size of ONE fp32 rounding of that product, which
// the fp32 kernels above commit as well. tests/test_proj_gpu.py pins it: against a float64 product the error of this path
// is not above that of the fp32-MFMA kernels (nor of torch's fp32 GEMM). Six bf16 MFMAs cost 6/16 of one fp32 MFMA per
// flop, so the matrix pipe is ~35 % busy and the launch is bound by the feature stream: HBM, where it belongs
// (SURVEY 7.2-8 allows exactly this: "no bf16 in the projection unless split-precision").
//
// Operands:
//   long operand  A [rows, red] fp32 in a TILE-MAJOR image made once (the features are constants, Models.py:46-47;
//                 mmssl_projx_pack_f32): the 256 x 32 slice (tile t, slice s) is one contiguous 32 KB block in the order
//                 the kernel's lanes consume it (see projx_pack_kernel), so every load instruction of a wave streams one
//                 contiguous KB straight into registers. Forward: A = F_g [M, K_g]; weight gradient: A = F_g^T
//                 [K_g, M] (a second packed copy: +1 x the feature bytes of HBM, irrelevant on 288 GB) - ONE kernel serves
//                 both directions. Zero-padded to whole tiles / slices.
//   short operand B [64, red] as three bf16 planes in the same slice-major form (12 KB per slice: plane p, row j = 64 B,
//                 8-value chunk c at position c ^ ((j >> 2) & 3)), produced per launch by a small kernel: the weights W_g
//                 (forward: projx_wsplit_kernel) or the transposed masked output gradient G^T (weight gradient:
//                 projx_gprep_kernel, which also leaves the bias-gradient column sums).
// The A values are cut into their three planes in registers (12 integer / fp32 VALU operations per pair of values, on the
// vector pipe beside the MFMAs); a wave loads exactly the 32 rows its MFMAs read.
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kXAFloats = PT * PBK;                    // 8192 floats = 32 KB: one A slice
constexpr int kXBBytes = 3 * PJ * PBK * 2;             // 12288 B: the three bf16 planes of one B slice
constexpr int kXGRows = 64;                            // reduction rows per block of the G preparation kernel

__device__ __forceinline__ unsigned f2u(float x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ float u2f(unsigned x) { return __builtin_bit_cast(float, x); }

// a = hi + mid + lo exactly, each piece a bf16 (the top 16 bits of an fp32): the bit patterns of the three pieces.
// FINITE inputs only: for a = +-Inf the remainder a - hi is NaN, so an infinite feature / weight / gradient gives NaN where
// the fp32-MFMA kernels (ops.PROJ_SPLIT = False) give Inf - both are a diverged run; NaN stays NaN. A remainder below the
// bf16 MFMA's denormal threshold may be flushed: it is < 2^-126, far below the 2^-24 relative error budget of the scheme.
__device__ __forceinline__ void cut3(float a, unsigned& hi, unsigned& mid, unsigned& lo) {
  hi = f2u(a) & 0xffff0000u;
  const float r = a - u2f(hi);            // exact: the low 16 significand bits
  mid = f2u(r) & 0xffff0000u;
  lo = f2u(r - u2f(mid));                 // <= 8 significant bits left: its top half is the whole value
}
// eight consecutive reduction values -> three packed bf16x8 operands (element e in bits 16 (e & 1) of dword e >> 1)
__device__ __forceinline__ void cut8(const float4& v0, const float4& v1, uintx4& H, uintx4& Mi, uintx4& L) {
  const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned h0, m0, l0, h1, m1, l1;
    cut3(x[2 * i], h0, m0, l0);
    cut3(x[2 * i + 1], h1, m1, l1);
    H[i] = h1 | (h0 >> 16);
    Mi[i] = m1 | (m0 >> 16);
    L[i] = (l1 & 0xffff0000u) | (l0 >> 16);
  }
}

struct FragX {
  uintx4 a[3][2];          // [plane][k-step]: this lane's row, reduction values 16 step + 8 h .. + 7
  uintx4 b[3][2][2];       // [plane][k-step][column half]: channel 32 half + (lane & 31), same reduction values
};

// 16 bytes global -> VGPRs, non-temporal. A PLAIN load: the kernel keeps kXDepth slices of its A operand in flight in
// registers, and the wait for them is the compiler's (SIInsertWaitcnts puts the vmcnt in front of the first reader, and in
// front of any copy the register allocator makes of such a register). Until round 6 these loads were inline asm waited for
// with the hand-counted vmcnt that B's LDS-DMA pieces need anyway: to the compiler the destination of an asm load is a
// value that exists as soon as the statement has executed, so it may copy it, or reuse a dead prefetch's register, before
// the hand-written wait - the cause of the round-5 intermittent weight gradient in linear.hip's wgrad10_kernel. The DMA
// pieces (asm, invisible to the compiler) sit between the A loads in the memory queue, so the compiler's count for A is
// never too lenient: it waits for at most two loads more than the hand count did.
typedef const __attribute__((address_space(1))) floatx4* gvec4;
typedef const __attribute__((address_space(1))) char* gbytes;
// a pointer every lane holds the same value of, as the SGPR pair the saddr forms want (the divergence analysis does not
// see through the segment lookup's VALU divisions)
__device__ __forceinline__ const char* uniform_ptr(const void* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}
// wave-uniform base + 32-bit per-lane byte offset (+ immediate): the saddr form of global_load
template <int OFF>
__device__ __forceinline__ void gload16_nt(floatx4& dst, const char* sbase, unsigned voff) {
  dst = __builtin_nontemporal_load((gvec4)((gbytes)(uintptr_t)sbase + voff + OFF));
}
// "the readers of these four registers start here": a side-effecting statement (ordered with the barriers and the DMA
// issues) that the cut of the slice into bf16 planes cannot be hoisted across; the compiler's vmcnt lands in front of it
__device__ __forceinline__ void tie4(floatx4& a, floatx4& b, floatx4& c, floatx4& d) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}

// slices in flight per wave (template parameter of the kernel): A in registers, B in the LDS ring of that many stages
// only the short operand passes through LDS; one stage more = the dump the issues past a range's end write into
constexpr int x_lds_bytes(int depth) { return (depth + 1) * kXBBytes; }

// The long operand never touches LDS: a wave's MFMA rows are read by that wave alone, so each lane loads the 16 values of
// its row straight into registers - four fully coalesced 1 KB loads per wave and slice out of an image laid out for
// exactly that (projx_pack_kernel) - kXDepth slices ahead (64 KB per CU in flight beside the B ring's DMA; the LDS-staged
// form of this kernel measured 92 us for the 387 MB stream: every byte paid the LDS-DMA path's ~6.4 TB/s chip-wide
// ceiling TOGETHER with the L2-resident B planes, +37 % bytes). LDS carries the three bf16 planes of the short operand
// only (12 KB per slice, kXDepth stages, one barrier per slice).
// Decomposition builds for tools/projx_ablate.sh (never set in the product build): bit 0 = no A loads in the steady loop,
// bit 1 = no MFMAs, bit 2 = no cut of A into planes, bit 3 = no B DMA in the steady loop.
#ifndef MMSSL_PROJX_DBG
#define MMSSL_PROJX_DBG 0
#endif
template <int kXDepth>
__global__ __launch_bounds__(kThreads) void projx_sk_kernel(Group P, int upb, int64_t total, int max_segs,
                                                            float* __restrict__ partials) {
  extern __shared__ __attribute__((aligned(16))) float ring[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, lr = lane & 31;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)ring);
  const int64_t u_begin = (int64_t)blockIdx.x * upb;
  const int64_t u_end = min(total, u_begin + upb);
  const int n = (int)(u_end - u_begin);          // slices of this block's range
  if (n <= 0) return;
  // the wave's two B pieces of a slice: KB wave and KB 8 + wave of the 12 (waves 4-7 repeat KB 0-3: the same bytes to the
  // same place, so that every wave has the same number of memory instructions in flight - the wait counts are immediates)
  const int be0 = wave_u, be1 = (8 + wave_u) % 12;
  // The block's whole range is ONE pipeline: the loads run kXDepth slices ahead of the MFMAs ACROSS segment boundaries (a
  // segment = the slices of one output tile; 2.5 per block on the Baby shape). Draining and refilling at every boundary,
  // as the fp32 kernels above do, cost ~5 us each with four slices in flight. Two cursors walk the range: the issue cursor
  // (source pointers of the next slice to load) and the compute cursor (slices left in the tile being accumulated).
  //
  // EVERY step issues the same memory instructions, on every path: past the end of the range the issues still go out,
  // from one lane-uniform line of the first image (a single 64-byte request) into registers nobody reads and into the
  // LDS dump stage. The memory queue then has ONE shape: the hand-counted waits for the DMA pieces are constants (no
  // ladder over "how many slices are left"), and the compiler's own vmcnt for the register loads is exact - a pass that
  // merges control-flow paths can only keep the smallest count, so with conditional issues it drained the ring
  // (vmcnt(0)) in every step. tools/vmcnt_check.py walks the same graph and needs the same property.
  // issue cursor: wave-uniform sources of the next slice's A values / B pieces (+ this lane's constant byte offsets)
  const char* pa = nullptr;
  const char* pb = nullptr;
  const unsigned la = (unsigned)(wave * 1024 + 4 * lane) * 4u, lb = 16u * (unsigned)lane;
  int i_left = 0;                  // slices left in the issue cursor's segment
  int i_k = 0;                     // slices issued so far (32-bit: `i_k < n` is a scalar compare, the select stays in SGPRs)
  floatx4 raw[kXDepth][4];
  const unsigned dump_lds = ring_lds + (unsigned)kXDepth * kXBBytes;
  // The issue cursor's next slice: `issue_B` sends its two DMA pieces of B into LDS stage k % depth, `issue_A` its four
  // loads of A into register set J and advances the cursor. B runs THREE slices ahead of the MFMAs, A four: a stage is
  // then refilled a whole step after its last operand read was issued, so the step's barrier only needs the reads
  // issued before the latest six (lgkmcnt(6)) - the reads of the slice's second half no longer stall the barrier.
  // Slices k >= n do not exist: their issues re-read the first line of the range's LAST slice (every lane the same 16
  // bytes) - no second pointer to carry, and the address is one this block has just read.
  auto open_segment = [&]() {
    if (i_left == 0) {
      const Segment sg = segment_at(P, u_begin + i_k, u_end);
      pa = uniform_ptr(P.A[sg.g] + ((int64_t)sg.tip * P.S[sg.g] + sg.s0) * kXAFloats);
      pb = uniform_ptr(reinterpret_cast<const char*>(P.B[sg.g]) + (int64_t)sg.s0 * kXBBytes);
      i_left = sg.nk;
    }
  };
  // B of slice k: called right BEFORE issue_A of the slice after it (B lags A by one slice), i.e. when `pb` has been
  // advanced past slice k (or past the last slice, for k >= n) and the next segment has not been opened yet
  auto issue_B = [&](int k) {
    const bool real = k < n;
    const unsigned st = real ? ring_lds + (unsigned)(k % kXDepth) * kXBBytes : dump_lds;
    const unsigned off = real ? lb : 0u;
    const char* src = pb - kXBBytes;
    if (!(MMSSL_PROJX_DBG & 8) || k < kXDepth) {
      glds16_s(src + be0 * 1024, off, st + (unsigned)(be0 * 1024));
      glds16_s(src + be1 * 1024, off, st + (unsigned)(be1 * 1024));
    }
  };
  auto issue_A = [&](auto J, int k) {
    const bool real = k < n;
    if (real) open_segment();
    const char* src = real ? pa : pa - kXAFloats * 4;
    const unsigned off = real ? la : 0u;
    if (!(MMSSL_PROJX_DBG & 1) || k < kXDepth) {
      gload16_nt<0>(raw[J.value][0], src, off);
      gload16_nt<1024>(raw[J.value][1], src, off);
      gload16_nt<2048>(raw[J.value][2], src, off);
      gload16_nt<3072>(raw[J.value][3], src, off);
    }
    if (real) {
      pa += kXAFloats * 4;
      pb += kXBBytes;
      --i_left;
      ++i_k;
    }
  };
  // the operands of k-step s of slice k: my A values from register set J (cut into planes here), six reads of B planes
  auto take_half = [&](auto J, int k, int s, FragX& f) {
    const char* st = reinterpret_cast<const char*>(ring) + (k % kXDepth) * kXBBytes;
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int nn = 32 * j + lr;
        f.b[p][s][j] = *reinterpret_cast<const uintx4*>(st + p * (PJ * PBK * 2) + nn * (PBK * 2) +
                                                        (((2 * s + h) ^ ((nn >> 2) & 3)) << 4));
      }
    const floatx4 v0 = raw[J.value][2 * s], v1 = raw[J.value][2 * s + 1];
    if (MMSSL_PROJX_DBG & 4) {         // decomposition build: the raw bits as "planes" (wrong numbers, same data flow)
      f.a[0][s] = __builtin_bit_cast(uintx4, v0);
      f.a[1][s] = __builtin_bit_cast(uintx4, v1);
      f.a[2][s] = __builtin_bit_cast(uintx4, v0);
    } else {
      cut8(make_float4(v0[0], v0[1], v0[2], v0[3]), make_float4(v1[0], v1[1], v1[2], v1[3]), f.a[0][s], f.a[1][s],
           f.a[2][s]);
    }
  };
  floatx16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  // k-step s: the six partial products of both column halves, smallest first, the two accumulators alternating
  auto mfma_half = [&](int s, const FragX& f) {
    constexpr int pa_[6] = {0, 2, 1, 0, 1, 0};
    constexpr int pb_[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      if (MMSSL_PROJX_DBG & 2) {         // decomposition build: keep the operands alive without the matrix pipe
        acc0[t] += u2f(f.a[pa_[t]][s][0] ^ f.b[pb_[t]][s][0][1]);
        acc1[t] += u2f(f.a[pa_[t]][s][2] ^ f.b[pb_[t]][s][1][3]);
        continue;
      }
      const bf16x8 av = __builtin_bit_cast(bf16x8, f.a[pa_[t]][s]);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, f.b[pb_[t]][s][0]), acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, f.b[pb_[t]][s][1]), acc1, 0, 0, 0);
    }
  };
  int c_left = segment_at(P, u_begin, u_end).nk;       // compute cursor: slices left in the tile being accumulated
  int64_t c_u = u_begin;
  int seg = 0;
  // One pipeline step = slice k's 24 MFMAs (operand set f). Before them: slice k + 1 has landed (mine: vmcnt; everyone's
  // B: barrier). B of slice k + 3 and A of slice k + 4 are issued into the stage / register set their predecessors have
  // left; after each k-step's MFMAs slice k + 1's operands replace the ones those MFMAs read. A tile's last slice then leaves the accumulator image in
  // the range's next partial slot (the forward kernel's plane order: the 32x32 C layout is dtype-independent).
  // Steps k >= n (the loop below runs whole groups of kXDepth) only issue: see the note on the memory queue above.
  auto step = [&](auto J, int k, FragX& f) {
    constexpr int JN = (J.value + 1) % kXDepth;
    const bool more1 = k + 1 < n;
    // memory instructions younger than B(k + 1) in issue order: A(k + 2), B(k + 2), A(k + 3) = 4 + 2 + 4, always.
    // Unconditional (also in the steps past the end, where nothing reads what landed): a wait that only some paths
    // take leaves, for the compiler, loads in flight into registers it reuses - it then waits again, conservatively.
    vm_wait_n<10>();
    tie4(raw[JN][0], raw[JN][1], raw[JN][2], raw[JN][3]);
    __builtin_amdgcn_s_waitcnt(0xC07F | (6 << 8));          // lgkmcnt(6): all operand reads but the latest six
    asm volatile("" ::: "memory");
    bare_barrier();
    issue_B(k + kXDepth - 1);
    issue_A(J, k + kXDepth);
    if (k >= n) return;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      mfma_half(s, f);
      if (more1) take_half(std::integral_constant<int, JN>{}, k + 1, s, f);
      __builtin_amdgcn_sched_barrier(0);
    }
    ++c_u;
    if (--c_left == 0) {
      float4* Pq = reinterpret_cast<float4*>(partials + ((size_t)blockIdx.x * max_segs + seg) * kSlotFloats);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Pq[q * kThreads + tid] = make_float4(acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]);
        Pq[(4 + q) * kThreads + tid] = make_float4(acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
      ++seg;
      if (more1) c_left = segment_at(P, c_u, u_end).nk;
    }
  };
  // prologue, in the steady order (B of a slice goes out right before A of the next one): A0 | B0 A1 | B1 A2 | B2 A3
  static_assert(kXDepth == 4, "the prologue and the wait counts are written for four slices in flight");
  issue_A(std::integral_constant<int, 0>{}, 0);
  issue_B(0);
  issue_A(std::integral_constant<int, 1>{}, 1);
  issue_B(1);
  issue_A(std::integral_constant<int, 2>{}, 2);
  issue_B(2);
  issue_A(std::integral_constant<int, 3>{}, 3);
  vm_wait_n<16>();     // A(0) and B(0) have landed: younger than B(0) are A1 B1 A2 B2 A3 = 4 + 2 + 4 + 2 + 4
  tie4(raw[0][0], raw[0][1], raw[0][2], raw[0][3]);
  bare_barrier();
  FragX f;
  take_half(std::integral_constant<int, 0>{}, 0, 0, f);
  take_half(std::integral_constant<int, 0>{}, 0, 1, f);
  for (int k = 0; k < n; k += kXDepth) {
    step(std::integral_constant<int, 0>{}, k, f);
    step(std::integral_constant<int, 1>{}, k + 1, f);
    step(std::integral_constant<int, 2>{}, k + 2, f);
    step(std::integral_constant<int, 3>{}, k + 3, f);
  }
  // the issues past the end are still in flight: their DMA pieces must not land in LDS the next block already owns
  vm_wait_n<0>();
}

// one 16-byte chunk of each plane: the eight values x[0..7] of row j, chunk c of slice-image `img`
__device__ __forceinline__ void store_planes(char* img, int j, int c, const float* x) {
  uintx4 H, Mi, L;
  cut8(make_float4(x[0], x[1], x[2], x[3]), make_float4(x[4], x[5], x[6], x[7]), H, Mi, L);
  char* at = img + j * (PBK * 2) + ((c ^ ((j >> 2) & 3)) << 4);
  *reinterpret_cast<uintx4*>(at) = H;
  *reinterpret_cast<uintx4*>(at + PJ * PBK * 2) = Mi;
  *reinterpret_cast<uintx4*>(at + 2 * PJ * PBK * 2) = L;
}

// forward: W_g [64, K] fp32 -> the B image of problem g (ceil(K / 32) slices, zeros past K). One thread = one chunk.
struct XSplit {
  const float* W[kMaxProb];
  char* img[kMaxProb];
  int K[kMaxProb];
  int unit0[kMaxProb + 1];      // first (slice, row, chunk) unit of problem g
  int n;
};
__global__ __launch_bounds__(256) void projx_wsplit_kernel(XSplit X) {
  const int uidx = blockIdx.x * 256 + threadIdx.x;
  if (uidx >= X.unit0[X.n]) return;
  int g = 0;
  while (g + 1 < X.n && uidx >= X.unit0[g + 1]) ++g;
  const int r = uidx - X.unit0[g];
  const int c = r & 3, j = (r >> 2) & 63, s = r >> 8;
  const int K = X.K[g], k0 = 32 * s + 8 * c;
  float x[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = (k0 + e < K) ? X.W[g][(int64_t)j * K + k0 + e] : 0.f;
  store_planes(X.img[g] + (int64_t)s * kXBBytes, j, c, x);
}

// weight gradient: G [M, ldg] fp32 (problem g = columns 64 g .. 64 g + 63) -> the B image of G_g^T (ceil(M / 32) slices,
// zeros past M) + this block's column sums. grid = (ceil(M / kXGRows), n_prob).
__global__ __launch_bounds__(256) void projx_gprep_kernel(const float* __restrict__ G, int64_t ldg, int64_t M, int n_slices,
                                                          char* __restrict__ img0, int64_t img_stride,
                                                          float* __restrict__ bpart) {
  __shared__ float tile[kXGRows][PJ + 1];
  const int tid = threadIdx.x, g = blockIdx.y;
  const int64_t m0 = (int64_t)blockIdx.x * kXGRows;
  for (int i = tid; i < kXGRows * (PJ / 4); i += 256) {
    const int r = i >> 4, c4 = i & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 + r < M) v = *reinterpret_cast<const float4*>(G + (m0 + r) * ldg + (int64_t)g * PJ + 4 * c4);
    tile[r][4 * c4] = v.x; tile[r][4 * c4 + 1] = v.y; tile[r][4 * c4 + 2] = v.z; tile[r][4 * c4 + 3] = v.w;
  }
  __syncthreads();
  char* img = img0 + (int64_t)g * img_stride;
  for (int uidx = tid; uidx < (kXGRows / PBK) * PJ * 4; uidx += 256) {
    const int c = uidx & 3, j = (uidx >> 2) & 63, sl = uidx >> 8;
    const int64_t s = m0 / PBK + sl;
    if (s >= n_slices) break;
    float x[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = tile[32 * sl + 8 * c + e][j];
    store_planes(img + s * kXBBytes, j, c, x);
  }
  if (bpart && tid < PJ) {
    float sum = 0.f;
    for (int r = 0; r < kXGRows; ++r) sum += tile[r][tid];
    bpart[((size_t)blockIdx.x * gridDim.y + g) * PJ + tid] = sum;
  }
}

// the tile-major image of A = F (transpose 0: rows = M, reduction = K) or A = F^T (transpose 1): one thread per 16 bytes.
// Block (tile t, slice s) = 2048 float4: wave w's KB q (0..3) holds, at lane l, the float4 of row 256 t + 32 w + (l & 31),
// reduction values 32 s + 16 (q >> 1) + 8 (l >> 5) + 4 (q & 1) .. + 3 - what projx_sk_kernel's lane l feeds the MFMA.
__global__ __launch_bounds__(256) void projx_pack_kernel(const float* __restrict__ F, int64_t M, int64_t K, int64_t ldf,
                                                         int transpose, int64_t n_slices, int64_t n_chunks,
                                                         float* __restrict__ out) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= n_chunks) return;
  const int l = (int)(o & 63), q = (int)((o >> 6) & 3), w = (int)((o >> 8) & 7);
  const int64_t ts = o >> 11, s = ts % n_slices, t = ts / n_slices;
  const int64_t row = t * PT + 32 * w + (l & 31), col = s * PBK + 16 * (q >> 1) + 8 * (l >> 5) + 4 * (q & 1);
  const int64_t rows = transpose ? K : M, red = transpose ? M : K;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (row < rows) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (col + e < red) v[e] = transpose ? F[(col + e) * ldf + row] : F[row * ldf + col + e];
  }
  reinterpret_cast<float4*>(out)[o] = make_float4(v[0], v[1], v[2], v[3]);
}

// weight-gradient epilogue on the forward-shaped slots: the tile is C[i = feature column][j = channel];
// gW[j][i .. i + 3] is one float4; bias gradient = the G preparation kernel's block sums in block order
// grid = tiles x 8 planes x kXRParts: a weight-gradient problem has few tiles (20 for the Baby shape) with ~13 slots each, so
// a plane is cut into kXRParts blocks of 128 threads - 640 blocks instead of 160 hide the slot loads' latency (the launch sits
// on the step's critical chain and runs beside the tables' AdamW launch there: 39 us as 160 blocks)
constexpr int kXRParts = 4;
constexpr int kXRThreads = kThreads / kXRParts;
__global__ __launch_bounds__(kXRThreads) void projx_wgrad_reduce_kernel(Group P, int upb, int max_segs,
                                                                        const float* __restrict__ partials,
                                                                        const float* __restrict__ bpart, int n_gblocks,
                                                                        WgradPtrs ptrs, AdamSlots ad) {
  __shared__ float sh[2];
  const int part = (int)blockIdx.x % kXRParts, bq = (int)blockIdx.x / kXRParts;
  const int tid = part * kXRThreads + (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;      // position inside the plane
  const int t = bq >> 3, q = bq & 7;
  const int g = prob_of_tile(P, t), tip = t - P.tile0[g];
  if (ad.state) {
    if (threadIdx.x == 0) {
      const float step = ad.pre_ticked ? fmaxf(ad.state[0], 1.0f) : ad.state[0] + 1.0f;
      sh[0] = ad.lr / (1.0f - __builtin_amdgcn_exp2f(step * ad.log2_beta1));
      sh[1] = sqrtf(1.0f - __builtin_amdgcn_exp2f(step * ad.log2_beta2));
    }
    __syncthreads();
  }
  const float decay = 1.0f - ad.lr * ad.wd;
  const float4 v = sum_slots(P, t, g, tip, q, upb, max_segs, partials, tid);
  const int col = 32 * (q >> 2) + (lane & 31);
  const int64_t i = (int64_t)tip * PT + 32 * wave + 8 * (q & 3) + 4 * (lane >> 5);
  const int64_t K = P.I[g];
  if (i < K) {
    const int64_t o = (int64_t)col * K + i;
    if (ptrs.gW[g]) *reinterpret_cast<float4*>(ptrs.gW[g] + o) = v;
    if (ad.state && ad.W[g]) {
      float4 pp = *reinterpret_cast<float4*>(ad.W[g] + o);
      float4 mm = *reinterpret_cast<float4*>(ad.mW[g] + o);
      float4 vv = *reinterpret_cast<float4*>(ad.vW[g] + o);
      adamw_update(pp.x, v.x, mm.x, vv.x, sh[0], sh[1], decay, ad.beta1, ad.beta2, ad.eps);
      adamw_update(pp.y, v.y, mm.y, vv.y, sh[0], sh[1], decay, ad.beta1, ad.beta2, ad.eps);
      adamw_update(pp.z, v.z, mm.z, vv.z, sh[0], sh[1], decay, ad.beta1, ad.beta2, ad.eps);
      adamw_update(pp.w, v.w, mm.w, vv.w, sh[0], sh[1], decay, ad.beta1, ad.beta2, ad.eps);
      *reinterpret_cast<float4*>(ad.W[g] + o) = pp;
      *reinterpret_cast<float4*>(ad.mW[g] + o) = mm;
      *reinterpret_cast<float4*>(ad.vW[g] + o) = vv;
      if (ad.wplanes[g]) {
        // W[col, i .. i + 3] = half a 16-byte chunk of store_planes' layout: slice i / 32, row `col`, chunk (i % 32) / 8
        // (swizzled by the row), bytes 8 * ((i / 4) & 1) .. + 7 of each plane; element e in bits 16 (e & 1) of dword e / 2
        unsigned h[4], m[4], l[4];
        cut3(pp.x, h[0], m[0], l[0]);
        cut3(pp.y, h[1], m[1], l[1]);
        cut3(pp.z, h[2], m[2], l[2]);
        cut3(pp.w, h[3], m[3], l[3]);
        char* at = ad.wplanes[g] + (i >> 5) * (int64_t)kXBBytes + col * (PBK * 2) +
                   (((int)((i & 31) >> 3) ^ ((col >> 2) & 3)) << 4) + 8 * (int)((i >> 2) & 1);
        *reinterpret_cast<uint2*>(at) = make_uint2(h[1] | (h[0] >> 16), h[3] | (h[2] >> 16));
        *reinterpret_cast<uint2*>(at + PJ * PBK * 2) = make_uint2(m[1] | (m[0] >> 16), m[3] | (m[2] >> 16));
        *reinterpret_cast<uint2*>(at + 2 * PJ * PBK * 2) =
            make_uint2((l[1] & 0xffff0000u) | (l[0] >> 16), (l[3] & 0xffff0000u) | (l[2] >> 16));
      }
    }
  }
  if (tip == 0 && q == 0 && tid < PJ && bpart && (ptrs.gb[g] || (ad.state && ad.b[g]))) {
    float s = 0.f;
    for (int b0 = 0; b0 < n_gblocks; b0 += 8) {
      float pv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) pv[k] = bpart[((size_t)min(b0 + k, n_gblocks - 1) * P.n + g) * PJ + tid];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (b0 + k < n_gblocks) s += pv[k];
    }
    if (ptrs.gb[g]) ptrs.gb[g][tid] = s;
    if (ad.state && ad.b[g]) {
      float pp = ad.b[g][tid], mm = ad.mb[g][tid], vv = ad.vb[g][tid];
      adamw_update(pp, s, mm, vv, sh[0], sh[1], decay, ad.beta1, ad.beta2, ad.eps);
      ad.b[g][tid] = pp;
      ad.mb[g][tid] = mm;
      ad.vb[g][tid] = vv;
    }
  }
}

// =====================================================================================================================
// host side
// =====================================================================================================================
struct Plan {
  Group P;
  int upb, blocks, max_segs, tiles;
  int64_t total;
};

int n_cus() {
  static int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
    return v;
  }();
  return n;
}

// tile axis extents I[g], reduction lengths R[g]; tile height pt, `per_cu` unit ranges per CU
bool make_plan(int n_prob, const int64_t* I, const int64_t* R, Plan& pl, int pt = PT, int per_cu = 1, int n_blocks = 0) {
  if (n_prob < 1 || n_prob > kMaxProb) return false;
  Group& P = pl.P;
  P.n = n_prob;
  int64_t units = 0;
  int tiles = 0, min_s = 1 << 30;
  for (int g = 0; g < n_prob; ++g) {
    if (I[g] <= 0 || R[g] <= 0) return false;
    P.I[g] = I[g];
    P.R[g] = R[g];
    P.S[g] = (int)((R[g] + PBK - 1) / PBK);
    const int tg = (int)((I[g] + pt - 1) / pt);
    P.unit0[g] = units;
    P.tile0[g] = tiles;
    units += (int64_t)tg * P.S[g];
    tiles += tg;
    min_s = std::min(min_s, P.S[g]);
  }
  for (int g = n_prob; g <= kMaxProb; ++g) {
    P.unit0[g] = units;
    P.tile0[g] = tiles;
  }
  for (int g = n_prob; g < kMaxProb; ++g) {
    P.A[g] = P.B[g] = nullptr;
    P.lda[g] = P.ldb[g] = P.I[g] = P.R[g] = 0;
    P.S[g] = 1;
  }
  if (units > (1ll << 40) || tiles > (1 << 20)) return false;
  const int64_t slots_ = n_blocks > 0 ? n_blocks : (int64_t)n_cus() * per_cu;
  int64_t upb = (units + slots_ - 1) / slots_;
  const int64_t floor_ = std::min<int64_t>(min_s, 8);       // a range is at least 8 slices deep (or one whole tile)
  if (upb < floor_) upb = floor_;
  pl.upb = (int)upb;
  pl.total = units;
  pl.blocks = (int)((units + upb - 1) / upb);
  pl.max_segs = (int)(upb / min_s) + 2;
  pl.tiles = tiles;
  return true;
}

size_t plan_ws_bytes(const Plan& pl, int slot_floats = kSlotFloats) {
  // [partial slots | bias partials]
  return (size_t)pl.blocks * pl.max_segs * (slot_floats + PJ) * sizeof(float) + 64;
}

int lds_ready() {
  static const int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(proj_fwd_sk_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kFLdsBytes) |
                        (int)hipFuncSetAttribute(reinterpret_cast<const void*>(proj_wgrad_sk_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
  return rc;
}

bool fwd_shape_ok(int n_prob, const int* K, int64_t M, int N) {
  if (n_prob < 1 || n_prob > kMaxProb || N != PJ || M <= 0) return false;
  for (int g = 0; g < n_prob; ++g)
    if (K[g] < PBK || K[g] % PBK != 0) return false;
  return true;
}
bool wgrad_shape_ok(int n_prob, const int* K, int64_t M, int N) {
  if (n_prob < 1 || n_prob > kMaxProb || N != PJ || M <= 0) return false;
  for (int g = 0; g < n_prob; ++g)
    if (K[g] < 4 || K[g] % 4 != 0) return false;
  return true;
}

}  // namespace

extern "C" int mmssl_proj_supported(int n_prob, const int* K, int64_t M, int N, int wgrad) {
  if (!K) return 0;
  return (wgrad ? wgrad_shape_ok(n_prob, K, M, N) : fwd_shape_ok(n_prob, K, M, N)) ? 1 : 0;
}

extern "C" size_t mmssl_proj_workspace_bytes(int n_prob, const int* K, int64_t M, int N, int wgrad) {
  if (!mmssl_proj_supported(n_prob, K, M, N, wgrad)) return 0;
  int64_t I[kMaxProb], R[kMaxProb];
  for (int g = 0; g < n_prob; ++g) {
    I[g] = wgrad ? K[g] : M;
    R[g] = wgrad ? M : K[g];
  }
  Plan pl;
  if (wgrad) return make_plan(n_prob, I, R, pl) ? plan_ws_bytes(pl) : 0;
  return make_plan(n_prob, I, R, pl, FPT, kFBlocksPerCU) ? plan_ws_bytes(pl, kFSlotFloats) : 0;
}

extern "C" int mmssl_proj_fwd_f32(int n_prob, const float* const* F, const float* const* W, const float* const* bias,
                                  const int* K, int64_t M, int N, const uint8_t* keep, uint8_t* keep_out,
                                  const uint64_t* rng_state, float p_drop, float scale, float* Y, int64_t ldy,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  if (!F || !W || !K || !Y || !workspace) return MMSSL_E_BADARG;
  if (!fwd_shape_ok(n_prob, K, M, N)) return MMSSL_E_UNSUPP;
  if (ldy < (int64_t)n_prob * N || (ldy & 3) || ((uintptr_t)Y & 15) || ((uintptr_t)workspace & 15)) return MMSSL_E_BADARG;
  if (keep && keep_out) return MMSSL_E_BADARG;
  if (keep_out && (!rng_state || !(p_drop >= 0.f && p_drop < 1.f))) return MMSSL_E_BADARG;
  int64_t I[kMaxProb], R[kMaxProb];
  for (int g = 0; g < n_prob; ++g) {
    if (!F[g] || !W[g] || (((uintptr_t)F[g] | (uintptr_t)W[g]) & 15)) return MMSSL_E_BADARG;
    I[g] = M;
    R[g] = K[g];
  }
  Plan pl;
  if (!make_plan(n_prob, I, R, pl, FPT, kFBlocksPerCU)) return MMSSL_E_UNSUPP;
  if (workspace_bytes < plan_ws_bytes(pl, kFSlotFloats)) return MMSSL_E_WORKSPACE;
  if (lds_ready() != 0) return MMSSL_E_UNSUPP;
  for (int g = 0; g < n_prob; ++g) {
    pl.P.A[g] = F[g];
    pl.P.B[g] = W[g];
    pl.P.lda[g] = K[g];
    pl.P.ldb[g] = K[g];
  }
  hipStream_t s = as_stream(stream);
  float* part = reinterpret_cast<float*>(workspace);
  FwdPtrs ptrs;
  for (int g = 0; g < kMaxProb; ++g) ptrs.bias[g] = (bias && g < n_prob) ? bias[g] : nullptr;
  hipLaunchKernelGGL(proj_fwd_sk_kernel, dim3((unsigned)pl.blocks), dim3(kFThreads), kFLdsBytes, s, pl.P, pl.upb, pl.total,
                     pl.max_segs, part);
  MMSSL_LAUNCH_CHECK();
  hipLaunchKernelGGL(proj_fwd_reduce_kernel, dim3((unsigned)pl.tiles * 8), dim3(kFThreads), 0, s, pl.P, pl.upb,
                     pl.max_segs, (const float*)part, M, Y, ldy, ptrs, keep, keep_out, rng_state, p_drop, scale);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

static int wgrad_impl(int n_prob, const float* G, int64_t ldg, const float* const* F, const int* K, int64_t M, int N,
                      float* const* gW, float* const* gb, const AdamSlots& ad, void* workspace, size_t workspace_bytes,
                      void* stream);

extern "C" int mmssl_proj_wgrad_f32(int n_prob, const float* G, int64_t ldg, const float* const* F, const int* K,
                                    int64_t M, int N, float* const* gW, float* const* gb, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  AdamSlots ad = {};
  return wgrad_impl(n_prob, G, ldg, F, K, M, N, gW, gb, ad, workspace, workspace_bytes, stream);
}

extern "C" int mmssl_proj_wgrad_adamw_f32(int n_prob, const float* G, int64_t ldg, const float* const* F, const int* K,
                                          int64_t M, int N, float* const* gW, float* const* gb, float* const* W,
                                          float* const* mW, float* const* vW, float* const* b, float* const* mb,
                                          float* const* vb, const float* state, float lr, float beta1, float beta2,
                                          float eps, float weight_decay, int pre_ticked, void* workspace,
                                          size_t workspace_bytes, void* stream) {
  if (!state || !W || !mW || !vW || n_prob < 1 || n_prob > kMaxProb) return MMSSL_E_BADARG;
  AdamSlots ad = {};
  for (int g = 0; g < n_prob; ++g) {
    if (!W[g] || !mW[g] || !vW[g]) return MMSSL_E_BADARG;
    if (((uintptr_t)W[g] | (uintptr_t)mW[g] | (uintptr_t)vW[g]) & 15) return MMSSL_E_BADARG;
    ad.W[g] = W[g];
    ad.mW[g] = mW[g];
    ad.vW[g] = vW[g];
    if (b && b[g]) {
      if (!mb || !vb || !mb[g] || !vb[g]) return MMSSL_E_BADARG;
      ad.b[g] = b[g];
      ad.mb[g] = mb[g];
      ad.vb[g] = vb[g];
    }
  }
  ad.state = state;
  ad.lr = lr; ad.beta1 = beta1; ad.beta2 = beta2; ad.eps = eps; ad.wd = weight_decay;
  ad.log2_beta1 = (float)std::log2((double)beta1);
  ad.log2_beta2 = (float)std::log2((double)beta2);
  ad.pre_ticked = pre_ticked ? 1 : 0;
  return wgrad_impl(n_prob, G, ldg, F, K, M, N, gW, gb, ad, workspace, workspace_bytes, stream);
}

static int wgrad_impl(int n_prob, const float* G, int64_t ldg, const float* const* F, const int* K, int64_t M, int N,
                      float* const* gW, float* const* gb, const AdamSlots& ad, void* workspace, size_t workspace_bytes,
                      void* stream) {
  if (!G || !F || !K || !workspace || (!gW && !ad.state)) return MMSSL_E_BADARG;
  if (!wgrad_shape_ok(n_prob, K, M, N)) return MMSSL_E_UNSUPP;
  if (ldg < (int64_t)n_prob * N || (ldg & 3) || ((uintptr_t)G & 15) || ((uintptr_t)workspace & 15)) return MMSSL_E_BADARG;
  int64_t I[kMaxProb], R[kMaxProb];
  for (int g = 0; g < n_prob; ++g) {
    float* gw = gW ? gW[g] : nullptr;
    if (!F[g] || (!gw && !ad.state) || (((uintptr_t)F[g] | (uintptr_t)gw) & 15)) return MMSSL_E_BADARG;
    I[g] = K[g];
    R[g] = M;
  }
  Plan pl;
  if (!make_plan(n_prob, I, R, pl)) return MMSSL_E_UNSUPP;
  if (workspace_bytes < plan_ws_bytes(pl)) return MMSSL_E_WORKSPACE;
  if (lds_ready() != 0) return MMSSL_E_UNSUPP;
  for (int g = 0; g < n_prob; ++g) {
    pl.P.A[g] = F[g];
    pl.P.B[g] = G + (int64_t)g * N;
    pl.P.lda[g] = K[g];
    pl.P.ldb[g] = ldg;
  }
  hipStream_t s = as_stream(stream);
  float* part = reinterpret_cast<float*>(workspace);
  const size_t slots = (size_t)pl.blocks * pl.max_segs;
  float* bpart = part + slots * kSlotFloats;
  WgradPtrs ptrs;
  bool any_b = false;
  for (int g = 0; g < kMaxProb; ++g) {
    ptrs.gW[g] = (g < n_prob && gW) ? gW[g] : nullptr;
    ptrs.gb[g] = (g < n_prob && gb) ? gb[g] : nullptr;
    any_b = any_b || ptrs.gb[g] != nullptr || (ad.state && g < n_prob && ad.b[g] != nullptr);
  }
  hipLaunchKernelGGL(proj_wgrad_sk_kernel, dim3((unsigned)pl.blocks), dim3(kThreads), kLdsBytes, s, pl.P, pl.upb,
                     pl.total, pl.max_segs, part, any_b ? bpart : (float*)nullptr);
  MMSSL_LAUNCH_CHECK();
  hipLaunchKernelGGL(proj_wgrad_reduce_kernel, dim3((unsigned)pl.tiles * 8), dim3(kThreads), 0, s, pl.P, pl.upb,
                     pl.max_segs, (const float*)part, (const float*)bpart, ptrs, ad);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

// =====================================================================================================================
// split-precision path: host side
// =====================================================================================================================
namespace {

int64_t x_slices(int64_t red) { return (red + PBK - 1) / PBK; }
int64_t x_tiles(int64_t rows) { return (rows + PT - 1) / PT; }
size_t x_align(size_t b) { return (b + 255) & ~(size_t)255; }
size_t x_slot_bytes(const Plan& pl) { return x_align((size_t)pl.blocks * pl.max_segs * kSlotFloats * sizeof(float)); }

bool x_shape_ok(int n_prob, const int* K, int64_t M, int N) {
  if (n_prob < 1 || n_prob > kMaxProb || N != PJ || M <= 0 || M > (1ll << 31)) return false;
  for (int g = 0; g < n_prob; ++g)
    if (K[g] < 4 || K[g] % 4 != 0) return false;
  return true;
}

// Prefetch depth: 4 slices per wave (16 KB of A in registers per wave, 48 KB of B planes in LDS). Measured on the Baby shape
// (profiles/r05/projx_depth.txt): depth 4 109 / 113 us forward / weight gradient with their epilogues, depth 6 112 / 117
// (230 VGPRs), depth 8 127 / 133 (256 VGPRs + scratch): the stream is not short of bytes in flight.
constexpr int kXDepthUsed = 4;
int x_lds_ready() {
  static const int rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(projx_sk_kernel<kXDepthUsed>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, x_lds_bytes(kXDepthUsed));
  return rc;
}
void x_launch_main(const Plan& pl, float* part, hipStream_t s) {
  hipLaunchKernelGGL(projx_sk_kernel<kXDepthUsed>, dim3((unsigned)pl.blocks), dim3(kThreads), x_lds_bytes(kXDepthUsed), s, pl.P,
                     pl.upb, pl.total, pl.max_segs, part);
}

// n_blocks: unit ranges = blocks of the launch (0 = one per CU). A step that runs other kernels beside the projection
// (hotpath.HotPathStep: the GCN chain on a side stream) asks for fewer - 13/16 of the CUs measured best there
// (profiles/r05/projx_blocks.txt): the launch is bound by the feature stream, which 13/16 of the CUs still saturate, and the
// kernels beside it stop starving (a side-stream SpMM under a full-width launch: 63 us instead of 11).
bool x_plan(int n_prob, const int* K, int64_t M, int wgrad, int n_blocks, Plan& pl) {
  int64_t I[kMaxProb], R[kMaxProb];
  for (int g = 0; g < n_prob; ++g) {
    I[g] = wgrad ? K[g] : M;
    R[g] = wgrad ? M : K[g];
  }
  if (n_blocks < 0 || n_blocks > 4096) return false;
  return make_plan(n_prob, I, R, pl, PT, 1, n_blocks);
}

int x_wgrad_impl(int n_prob, const float* G, int64_t ldg, const float* const* FTimg, const int* K, int64_t M, int N,
                 float* const* gW, float* const* gb, const AdamSlots& ad, int n_blocks, void* workspace,
                 size_t workspace_bytes, void* stream) {
  if (!G || !FTimg || !K || !workspace || (!gW && !ad.state)) return MMSSL_E_BADARG;
  if (!x_shape_ok(n_prob, K, M, N)) return MMSSL_E_UNSUPP;
  if (ldg < (int64_t)n_prob * N || (ldg & 3) || ((uintptr_t)G & 15) || ((uintptr_t)workspace & 255)) return MMSSL_E_BADARG;
  Plan pl;
  if (!x_plan(n_prob, K, M, 1, n_blocks, pl)) return MMSSL_E_UNSUPP;
  if (workspace_bytes < mmssl_projx_workspace_bytes(n_prob, K, M, N, 1, n_blocks)) return MMSSL_E_WORKSPACE;
  if (x_lds_ready() != 0) return MMSSL_E_UNSUPP;
  const int64_t S = x_slices(M);
  const int gblocks = (int)((M + kXGRows - 1) / kXGRows);
  char* base = reinterpret_cast<char*>(workspace);
  float* part = reinterpret_cast<float*>(base);
  char* gimg = base + x_slot_bytes(pl);
  const int64_t img_stride = S * kXBBytes;
  float* bpart = reinterpret_cast<float*>(gimg + x_align((size_t)n_prob * img_stride));
  WgradPtrs ptrs;
  bool any_b = false;
  for (int g = 0; g < kMaxProb; ++g) {
    ptrs.gW[g] = (g < n_prob && gW) ? gW[g] : nullptr;
    ptrs.gb[g] = (g < n_prob && gb) ? gb[g] : nullptr;
    any_b = any_b || ptrs.gb[g] != nullptr || (ad.state && g < n_prob && ad.b[g] != nullptr);
    if (g < n_prob) {
      float* gw = gW ? gW[g] : nullptr;
      if (!FTimg[g] || (!gw && !ad.state) || (((uintptr_t)FTimg[g] | (uintptr_t)gw) & 15)) return MMSSL_E_BADARG;
      pl.P.A[g] = FTimg[g];
      pl.P.B[g] = reinterpret_cast<const float*>(gimg + (int64_t)g * img_stride);
      pl.P.lda[g] = pl.P.ldb[g] = 0;
    }
  }
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(projx_gprep_kernel, dim3((unsigned)gblocks, (unsigned)n_prob), dim3(256), 0, s, G, ldg, M, (int)S, gimg,
                     img_stride, any_b ? bpart : (float*)nullptr);
  MMSSL_LAUNCH_CHECK();
  x_launch_main(pl, part, s);
  MMSSL_LAUNCH_CHECK();
  hipLaunchKernelGGL(projx_wgrad_reduce_kernel, dim3((unsigned)pl.tiles * 8 * kXRParts), dim3(kXRThreads), 0, s, pl.P, pl.upb,
                     pl.max_segs,
                     (const float*)part, any_b ? (const float*)bpart : (const float*)nullptr, gblocks, ptrs, ad);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int mmssl_projx_supported(int n_prob, const int* K, int64_t M, int N) {
  return (K && x_shape_ok(n_prob, K, M, N)) ? 1 : 0;
}

extern "C" size_t mmssl_projx_image_floats(int64_t rows, int64_t red) {
  if (rows <= 0 || red <= 0) return 0;
  return (size_t)(x_tiles(rows) * x_slices(red)) * kXAFloats;
}

extern "C" int mmssl_projx_pack_f32(const float* F, int64_t M, int64_t K, int64_t ldf, int transpose, float* out,
                                    void* stream) {
  if (!F || !out || M <= 0 || K <= 0 || ldf < K || ((uintptr_t)out & 15)) return MMSSL_E_BADARG;
  const int64_t rows = transpose ? K : M, red = transpose ? M : K;
  const int64_t S = x_slices(red), chunks = x_tiles(rows) * S * (kXAFloats / 4);
  hipLaunchKernelGGL(projx_pack_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, as_stream(stream), F, M, K, ldf,
                     transpose ? 1 : 0, S, chunks, out);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t mmssl_projx_workspace_bytes(int n_prob, const int* K, int64_t M, int N, int wgrad, int n_blocks) {
  if (!K || !x_shape_ok(n_prob, K, M, N)) return 0;
  Plan pl;
  if (!x_plan(n_prob, K, M, wgrad, n_blocks, pl)) return 0;
  size_t b = x_slot_bytes(pl);
  if (wgrad) {
    b += x_align((size_t)n_prob * x_slices(M) * kXBBytes);
    b += x_align((size_t)((M + kXGRows - 1) / kXGRows) * n_prob * PJ * sizeof(float));
  } else {
    for (int g = 0; g < n_prob; ++g) b += x_align((size_t)x_slices(K[g]) * kXBBytes);
  }
  return b + 256;
}

namespace {
// the bf16 planes of the weights (the short operand of the forward): problem g's image at wimg + off[g]
size_t x_wimg_layout(int n_prob, const int* K, size_t* off) {
  size_t o = 0;
  for (int g = 0; g < n_prob; ++g) {
    if (off) off[g] = o;
    o += x_align((size_t)x_slices(K[g]) * kXBBytes);
  }
  return o;
}
int x_wsplit(int n_prob, const float* const* W, const int* K, char* wimg, hipStream_t s) {
  XSplit X;
  X.n = n_prob;
  size_t off[kMaxProb];
  x_wimg_layout(n_prob, K, off);
  int units = 0;
  for (int g = 0; g < kMaxProb; ++g) {
    X.unit0[g] = units;
    X.W[g] = nullptr; X.img[g] = nullptr; X.K[g] = 0;
    if (g < n_prob) {
      if (!W[g] || ((uintptr_t)W[g] & 15)) return MMSSL_E_BADARG;
      X.W[g] = W[g];
      X.img[g] = wimg + off[g];
      X.K[g] = K[g];
      units += (int)x_slices(K[g]) * PJ * 4;
    }
  }
  X.unit0[kMaxProb] = units;
  for (int g = n_prob; g <= kMaxProb; ++g) X.unit0[g] = units;
  hipLaunchKernelGGL(projx_wsplit_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, X);
  MMSSL_LAUNCH_CHECK();
  return 0;
}
// W != NULL: the planes are made here, into the workspace (one more launch in front of the main kernel); W == NULL: `wimg`
// holds them already (mmssl_projx_wsplit_f32 by the caller, e.g. right behind the optimiser's update of the weights)
int x_fwd_impl(int n_prob, const float* const* Fimg, const float* const* W, const void* wimg_in, const float* const* bias,
               const int* K, int64_t M, int N, const uint8_t* keep, uint8_t* keep_out, const uint64_t* rng_state,
               float p_drop, float scale, float* Y, int64_t ldy, int n_blocks, void* workspace, size_t workspace_bytes,
               void* stream) {
  if (!Fimg || (!W && !wimg_in) || !K || !Y || !workspace) return MMSSL_E_BADARG;
  if (!x_shape_ok(n_prob, K, M, N)) return MMSSL_E_UNSUPP;
  if (ldy < (int64_t)n_prob * N || (ldy & 3) || ((uintptr_t)Y & 15) || ((uintptr_t)workspace & 255)) return MMSSL_E_BADARG;
  if (wimg_in && ((uintptr_t)wimg_in & 255)) return MMSSL_E_BADARG;
  if (keep && keep_out) return MMSSL_E_BADARG;
  if (keep_out && (!rng_state || !(p_drop >= 0.f && p_drop < 1.f))) return MMSSL_E_BADARG;
  Plan pl;
  if (!x_plan(n_prob, K, M, 0, n_blocks, pl)) return MMSSL_E_UNSUPP;
  if (workspace_bytes < mmssl_projx_workspace_bytes(n_prob, K, M, N, 0, n_blocks)) return MMSSL_E_WORKSPACE;
  if (x_lds_ready() != 0) return MMSSL_E_UNSUPP;
  char* base = reinterpret_cast<char*>(workspace);
  float* part = reinterpret_cast<float*>(base);
  char* wimg = wimg_in ? const_cast<char*>(reinterpret_cast<const char*>(wimg_in)) : base + x_slot_bytes(pl);
  size_t off[kMaxProb];
  x_wimg_layout(n_prob, K, off);
  for (int g = 0; g < n_prob; ++g) {
    if (!Fimg[g] || ((uintptr_t)Fimg[g] & 15)) return MMSSL_E_BADARG;
    pl.P.A[g] = Fimg[g];
    pl.P.B[g] = reinterpret_cast<const float*>(wimg + off[g]);
    pl.P.lda[g] = pl.P.ldb[g] = 0;
  }
  hipStream_t s = as_stream(stream);
  FwdPtrs ptrs;
  for (int g = 0; g < kMaxProb; ++g) ptrs.bias[g] = (bias && g < n_prob) ? bias[g] : nullptr;
  if (!wimg_in) {
    const int rc = x_wsplit(n_prob, W, K, wimg, s);
    if (rc) return rc;
  }
  x_launch_main(pl, part, s);
  MMSSL_LAUNCH_CHECK();
  hipLaunchKernelGGL(proj_fwd_reduce_kernel, dim3((unsigned)pl.tiles * 8), dim3(kFThreads), 0, s, pl.P, pl.upb, pl.max_segs,
                     (const float*)part, M, Y, ldy, ptrs, keep, keep_out, rng_state, p_drop, scale);
  MMSSL_LAUNCH_CHECK();
  return 0;
}
}  // namespace

extern "C" int mmssl_projx_fwd_f32(int n_prob, const float* const* Fimg, const float* const* W, const float* const* bias,
                                   const int* K, int64_t M, int N, const uint8_t* keep, uint8_t* keep_out,
                                   const uint64_t* rng_state, float p_drop, float scale, float* Y, int64_t ldy,
                                   int n_blocks, void* workspace, size_t workspace_bytes, void* stream) {
  if (!W) return MMSSL_E_BADARG;
  return x_fwd_impl(n_prob, Fimg, W, nullptr, bias, K, M, N, keep, keep_out, rng_state, p_drop, scale, Y, ldy, n_blocks,
                    workspace, workspace_bytes, stream);
}

extern "C" size_t mmssl_projx_wimg_bytes(int n_prob, const int* K) {
  if (!K || n_prob < 1 || n_prob > kMaxProb) return 0;
  for (int g = 0; g < n_prob; ++g)
    if (K[g] < 4 || K[g] % 4 != 0) return 0;
  return x_wimg_layout(n_prob, K, nullptr);
}

extern "C" int mmssl_projx_wsplit_f32(int n_prob, const float* const* W, const int* K, void* wimg, void* stream) {
  if (!W || !K || !wimg || ((uintptr_t)wimg & 255) || mmssl_projx_wimg_bytes(n_prob, K) == 0) return MMSSL_E_BADARG;
  return x_wsplit(n_prob, W, K, reinterpret_cast<char*>(wimg), as_stream(stream));
}

extern "C" int mmssl_projx_fwd_img_f32(int n_prob, const float* const* Fimg, const void* wimg, const float* const* bias,
                                       const int* K, int64_t M, int N, const uint8_t* keep, uint8_t* keep_out,
                                       const uint64_t* rng_state, float p_drop, float scale, float* Y, int64_t ldy,
                                       int n_blocks, void* workspace, size_t workspace_bytes, void* stream) {
  if (!wimg) return MMSSL_E_BADARG;
  return x_fwd_impl(n_prob, Fimg, nullptr, wimg, bias, K, M, N, keep, keep_out, rng_state, p_drop, scale, Y, ldy, n_blocks,
                    workspace, workspace_bytes, stream);
}

extern "C" int mmssl_projx_wgrad_f32(int n_prob, const float* G, int64_t ldg, const float* const* FTimg, const int* K,
                                     int64_t M, int N, float* const* gW, float* const* gb, int n_blocks, void* workspace,
                                     size_t workspace_bytes, void* stream) {
  AdamSlots ad = {};
  return x_wgrad_impl(n_prob, G, ldg, FTimg, K, M, N, gW, gb, ad, n_blocks, workspace, workspace_bytes, stream);
}

extern "C" int mmssl_projx_wgrad_adamw_img_f32(int n_prob, const float* G, int64_t ldg, const float* const* FTimg,
                                               const int* K, int64_t M, int N, float* const* gW, float* const* gb,
                                               float* const* W, float* const* mW, float* const* vW, float* const* b,
                                               float* const* mb, float* const* vb, const float* state, float lr, float beta1,
                                               float beta2, float eps, float weight_decay, int pre_ticked, void* wimg,
                                               int n_blocks, void* workspace, size_t workspace_bytes, void* stream);

extern "C" int mmssl_projx_wgrad_adamw_f32(int n_prob, const float* G, int64_t ldg, const float* const* FTimg, const int* K,
                                           int64_t M, int N, float* const* gW, float* const* gb, float* const* W,
                                           float* const* mW, float* const* vW, float* const* b, float* const* mb,
                                           float* const* vb, const float* state, float lr, float beta1, float beta2,
                                           float eps, float weight_decay, int pre_ticked, int n_blocks, void* workspace,
                                           size_t workspace_bytes, void* stream) {
  return mmssl_projx_wgrad_adamw_img_f32(n_prob, G, ldg, FTimg, K, M, N, gW, gb, W, mW, vW, b, mb, vb, state, lr, beta1, beta2,
                                         eps, weight_decay, pre_ticked, nullptr, n_blocks, workspace, workspace_bytes, stream);
}

// `wimg` (may be NULL): a mmssl_projx_wsplit_f32 image of these weights; the epilogue rewrites the planes of every weight it
// updates (the zero padding past K is never touched), so the image stays that of the CURRENT weights without a split launch
extern "C" int mmssl_projx_wgrad_adamw_img_f32(int n_prob, const float* G, int64_t ldg, const float* const* FTimg,
                                               const int* K, int64_t M, int N, float* const* gW, float* const* gb,
                                               float* const* W, float* const* mW, float* const* vW, float* const* b,
                                               float* const* mb, float* const* vb, const float* state, float lr, float beta1,
                                               float beta2, float eps, float weight_decay, int pre_ticked, void* wimg,
                                               int n_blocks, void* workspace, size_t workspace_bytes, void* stream) {
  if (!state || !W || !mW || !vW || !K || n_prob < 1 || n_prob > kMaxProb) return MMSSL_E_BADARG;
  AdamSlots ad = {};
  if (wimg) {
    if ((uintptr_t)wimg & 255) return MMSSL_E_BADARG;
    size_t off[kMaxProb];
    if (x_wimg_layout(n_prob, K, off) == 0) return MMSSL_E_BADARG;
    for (int g = 0; g < n_prob; ++g) ad.wplanes[g] = static_cast<char*>(wimg) + off[g];
  }
  for (int g = 0; g < n_prob; ++g) {
    if (!W[g] || !mW[g] || !vW[g]) return MMSSL_E_BADARG;
    if (((uintptr_t)W[g] | (uintptr_t)mW[g] | (uintptr_t)vW[g]) & 15) return MMSSL_E_BADARG;
    ad.W[g] = W[g];
    ad.mW[g] = mW[g];
    ad.vW[g] = vW[g];
    if (b && b[g]) {
      if (!mb || !vb || !mb[g] || !vb[g]) return MMSSL_E_BADARG;
      ad.b[g] = b[g];
      ad.mb[g] = mb[g];
      ad.vb[g] = vb[g];
    }
  }
  ad.state = state;
  ad.lr = lr; ad.beta1 = beta1; ad.beta2 = beta2; ad.eps = eps; ad.wd = weight_decay;
  ad.log2_beta1 = (float)std::log2((double)beta1);
  ad.log2_beta2 = (float)std::log2((double)beta2);
  ad.pre_ticked = pre_ticked ? 1 : 0;
  return x_wgrad_impl(n_prob, G, ldg, FTimg, K, M, N, gW, gb, ad, n_blocks, workspace, workspace_bytes, stream);
}
