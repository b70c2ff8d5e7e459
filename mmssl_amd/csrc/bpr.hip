// Fused BPR loss: gather + dot + logsigmoid + L2 regulariser, forward and backward.
// Replaces the three index gathers and Trainer.bpr_loss
// (/root/reference/MMSSL/main.py:368-371, 499-511).
//
// One lane group (d/4 lanes, a float4 each) per sample: the three rows are read as single
// coalesced requests; the per-sample dot products and squared norms are wavefront shuffle
// reductions; block partials are combined in a fixed order (deterministic loss).
// The backward scatters into the dense table gradients with hardware fp32 atomics (only
// rows that repeat inside a batch — possible for items — see their add order vary).
#include "common.hpp"
#include "bpr_step.hpp"

using namespace mmssl;

namespace {

struct BprRows {
  const float4* u;
  const float4* p;
  const float4* n;
};

template <int LPR>
__device__ __forceinline__ void bpr_rows(const float4* Eu, const float4* Ei, const float4* Ein,
                                         const int64_t* users, const int64_t* pos, const int64_t* neg,
                                         int64_t b, int lig, float4& u, float4& p, float4& n,
                                         int64_t& ru, int64_t& rp, int64_t& rn) {
  if (users) {
    ru = users[b]; rp = pos[b]; rn = neg[b];
    u = Eu[ru * LPR + lig];
    p = Ei[rp * LPR + lig];
    n = Ei[rn * LPR + lig];
  } else {
    ru = rp = rn = b;
    u = Eu[b * LPR + lig];
    p = Ei[b * LPR + lig];
    n = Ein[b * LPR + lig];
  }
}

__device__ __forceinline__ float log_sigmoid(float x) {
  // min(x,0) - log1p(exp(-|x|))  (what F.logsigmoid computes)
  return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}

// part[2*block + {0,1}] = { sum logsigmoid(diff), sum (|u|^2+|p|^2+|n|^2) } over the block's samples
template <int LPR>
__global__ __launch_bounds__(kBlock) void bpr_fwd_kernel(const float4* __restrict__ Eu,
                                                         const float4* __restrict__ Ei,
                                                         const float4* __restrict__ Ein,
                                                         const int64_t* __restrict__ users,
                                                         const int64_t* __restrict__ pos,
                                                         const int64_t* __restrict__ neg, int64_t B,
                                                         float* __restrict__ part) {
  __shared__ float red[4];
  constexpr int GPB = kBlock / LPR;
  const int lig = threadIdx.x & (LPR - 1);
  const int64_t b = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
  float ls = 0.f, sq = 0.f;
  if (b < B) {
    float4 u, p, n;
    int64_t ru, rp, rn;
    bpr_rows<LPR>(Eu, Ei, Ein, users, pos, neg, b, lig, u, p, n, ru, rp, rn);
    const float sp = group_sum<LPR>(f4_dot(u, p));
    const float sn = group_sum<LPR>(f4_dot(u, n));
    const float q = group_sum<LPR>(f4_dot(u, u) + f4_dot(p, p) + f4_dot(n, n));
    if (lig == 0) {
      ls = log_sigmoid(sp - sn);
      sq = q;
    }
  }
  const float t0 = block_sum_256(ls, red);
  const float t1 = block_sum_256(sq, red);
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x + 0] = t0;
    part[2 * blockIdx.x + 1] = t1;
  }
}

__global__ __launch_bounds__(kBlock) void bpr_finalize_kernel(const float* __restrict__ part, int nparts,
                                                              int64_t B, float decay, int64_t batch_size,
                                                              float* __restrict__ out3) {
  __shared__ float red[4];
  float a = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < nparts; i += kBlock) {
    a += part[2 * i + 0];
    c += part[2 * i + 1];
  }
  const float ls = block_sum_256(a, red);
  const float sq = block_sum_256(c, red);
  if (threadIdx.x == 0) {
    out3[0] = -(ls / (float)B);                              // mf_loss = -mean(logsigmoid)
    out3[1] = decay * ((0.5f * sq) / (float)batch_size);     // emb_loss
    out3[2] = 0.f;                                           // reg_loss (main.py:510)
  }
}

template <int LPR>
__global__ __launch_bounds__(kBlock) void bpr_bwd_kernel(const float4* __restrict__ Eu,
                                                         const float4* __restrict__ Ei,
                                                         const float4* __restrict__ Ein,
                                                         const int64_t* __restrict__ users,
                                                         const int64_t* __restrict__ pos,
                                                         const int64_t* __restrict__ neg, int64_t B,
                                                         float decay, int64_t batch_size,
                                                         const float* __restrict__ g_mf,
                                                         const float* __restrict__ g_emb,
                                                         float* __restrict__ gEu, float* __restrict__ gEi,
                                                         float* __restrict__ gEin) {
  constexpr int GPB = kBlock / LPR;
  const int lig = threadIdx.x & (LPR - 1);
  const int64_t b = (int64_t)blockIdx.x * GPB + threadIdx.x / LPR;
  if (b >= B) return;
  float4 u, p, n;
  int64_t ru, rp, rn;
  bpr_rows<LPR>(Eu, Ei, Ein, users, pos, neg, b, lig, u, p, n, ru, rp, rn);
  const float diff = group_sum<LPR>(f4_dot(u, p)) - group_sum<LPR>(f4_dot(u, n));
  // d(-mean logsigmoid(x))/dx = -sigmoid(-x)/B
  const float sig_neg = 1.f / (1.f + expf(diff));
  const float cm = -g_mf[0] * sig_neg / (float)B;
  const float ce = g_emb[0] * decay / (float)batch_size;
  const float4 gu = make_float4(cm * (p.x - n.x) + ce * u.x, cm * (p.y - n.y) + ce * u.y,
                                cm * (p.z - n.z) + ce * u.z, cm * (p.w - n.w) + ce * u.w);
  const float4 gp = make_float4(cm * u.x + ce * p.x, cm * u.y + ce * p.y, cm * u.z + ce * p.z,
                                cm * u.w + ce * p.w);
  const float4 gn = make_float4(-cm * u.x + ce * n.x, -cm * u.y + ce * n.y, -cm * u.z + ce * n.z,
                                -cm * u.w + ce * n.w);
  constexpr int D = LPR * 4;
  if (users) {
    float* du = gEu + ru * D + lig * 4;
    float* dp = gEi + rp * D + lig * 4;
    float* dn = gEi + rn * D + lig * 4;
    unsafeAtomicAdd(du + 0, gu.x); unsafeAtomicAdd(du + 1, gu.y); unsafeAtomicAdd(du + 2, gu.z); unsafeAtomicAdd(du + 3, gu.w);
    unsafeAtomicAdd(dp + 0, gp.x); unsafeAtomicAdd(dp + 1, gp.y); unsafeAtomicAdd(dp + 2, gp.z); unsafeAtomicAdd(dp + 3, gp.w);
    unsafeAtomicAdd(dn + 0, gn.x); unsafeAtomicAdd(dn + 1, gn.y); unsafeAtomicAdd(dn + 2, gn.z); unsafeAtomicAdd(dn + 3, gn.w);
  } else {
    reinterpret_cast<float4*>(gEu)[b * LPR + lig] = gu;
    reinterpret_cast<float4*>(gEi)[b * LPR + lig] = gp;
    reinterpret_cast<float4*>(gEin)[b * LPR + lig] = gn;
  }
}

// The hot step's loss tail as ONE launch (gather mode only): BPR backward for known upstream gradients (g_mf, g_emb
// on the device) + the BPR loss partials of the same gathered rows; the LAST block to arrive reduces the partials in
// block order (same arithmetic as bpr_finalize_kernel), writes terms[0..2], assembles
//   total = sum_k w[k] * terms[k] + c * extra        (main.py:420; terms[3..] are read: the InfoNCE losses)
// and advances the step's counters. `ticket` must be 0 on entry and is left 0.
template <int LPR>
__global__ __launch_bounds__(kBlock) void bpr_step_kernel(BprStepArgs A) {
  bpr_step_block<LPR>(A, (int)blockIdx.x);
}

inline int64_t bpr_blocks(int64_t B, int d) {
  const int gpb = kBlock / (d / 4);
  return (B + gpb - 1) / gpb;
}

}  // namespace

extern "C" size_t mmssl_bpr_workspace_bytes(int64_t B) {
  if (B <= 0) return 16;
  // enough for the smallest lane group (d=256: 4 samples per block)
  return (size_t)((B + 3) / 4) * 2 * sizeof(float) + 16;
}

extern "C" int mmssl_bpr_fwd_f32(const float* Eu, const float* Ei, const float* Ei_neg, const int64_t* users,
                                 const int64_t* pos, const int64_t* neg, int64_t B, int d, float decay,
                                 int64_t batch_size, float* out3, void* workspace, size_t workspace_bytes,
                                 void* stream) {
  if (B <= 0 || batch_size <= 0 || !Eu || !Ei || !out3) return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  const bool gather = users != nullptr;
  if (gather ? (!pos || !neg) : (pos || neg || !Ei_neg)) return MMSSL_E_BADARG;
  if (!workspace || workspace_bytes < mmssl_bpr_workspace_bytes(B)) return MMSSL_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  const int nb = (int)bpr_blocks(B, d);
  float* part = reinterpret_cast<float*>(workspace);
  const float4* a = reinterpret_cast<const float4*>(Eu);
  const float4* b = reinterpret_cast<const float4*>(Ei);
  const float4* c = reinterpret_cast<const float4*>(Ei_neg);
  switch (d) {
    case 32: hipLaunchKernelGGL((bpr_fwd_kernel<8>), dim3(nb), dim3(kBlock), 0, s, a, b, c, users, pos, neg, B, part); break;
    case 64: hipLaunchKernelGGL((bpr_fwd_kernel<16>), dim3(nb), dim3(kBlock), 0, s, a, b, c, users, pos, neg, B, part); break;
    case 128: hipLaunchKernelGGL((bpr_fwd_kernel<32>), dim3(nb), dim3(kBlock), 0, s, a, b, c, users, pos, neg, B, part); break;
    case 256: hipLaunchKernelGGL((bpr_fwd_kernel<64>), dim3(nb), dim3(kBlock), 0, s, a, b, c, users, pos, neg, B, part); break;
  }
  MMSSL_LAUNCH_CHECK();
  hipLaunchKernelGGL(bpr_finalize_kernel, dim3(1), dim3(kBlock), 0, s, part, nb, B, decay, batch_size, out3);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_bpr_bwd_f32(const float* Eu, const float* Ei, const float* Ei_neg, const int64_t* users,
                                 const int64_t* pos, const int64_t* neg, int64_t B, int d, float decay,
                                 int64_t batch_size, const float* g_mf, const float* g_emb, float* gEu,
                                 float* gEi, float* gEi_neg, void* stream) {
  if (B <= 0 || batch_size <= 0 || !Eu || !Ei || !g_mf || !g_emb || !gEu || !gEi) return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  const bool gather = users != nullptr;
  if (gather ? (!pos || !neg) : (pos || neg || !Ei_neg || !gEi_neg)) return MMSSL_E_BADARG;
  hipStream_t s = as_stream(stream);
  const int nb = (int)bpr_blocks(B, d);
  const float4* a = reinterpret_cast<const float4*>(Eu);
  const float4* b = reinterpret_cast<const float4*>(Ei);
  const float4* c = reinterpret_cast<const float4*>(Ei_neg);
  switch (d) {
    case 32: hipLaunchKernelGGL((bpr_bwd_kernel<8>), dim3(nb), dim3(kBlock), 0, s, a, b, c, users, pos, neg, B, decay, batch_size, g_mf, g_emb, gEu, gEi, gEi_neg); break;
    case 64: hipLaunchKernelGGL((bpr_bwd_kernel<16>), dim3(nb), dim3(kBlock), 0, s, a, b, c, users, pos, neg, B, decay, batch_size, g_mf, g_emb, gEu, gEi, gEi_neg); break;
    case 128: hipLaunchKernelGGL((bpr_bwd_kernel<32>), dim3(nb), dim3(kBlock), 0, s, a, b, c, users, pos, neg, B, decay, batch_size, g_mf, g_emb, gEu, gEi, gEi_neg); break;
    case 256: hipLaunchKernelGGL((bpr_bwd_kernel<64>), dim3(nb), dim3(kBlock), 0, s, a, b, c, users, pos, neg, B, decay, batch_size, g_mf, g_emb, gEu, gEi, gEi_neg); break;
  }
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_bpr_step_f32(const float* Eu, const float* Ei, const int64_t* users, const int64_t* pos,
                                  const int64_t* neg, int64_t B, int d, float decay, int64_t batch_size,
                                  const float* g_mf, const float* g_emb, float* gEu, float* gEi, float* terms,
                                  const float* w, int n_terms, const float* extra, float c, float* total,
                                  float* const* f32_ticks, int n_f32, uint64_t* const* u64_ticks, int n_u64,
                                  void* workspace, size_t workspace_bytes, int* ticket, const float* extra_parts,
                                  int64_t n_extra_parts, void* stream) {
  BprStepArgs A;
  const int rc = make_bpr_step_args(A, Eu, Ei, users, pos, neg, B, d, decay, batch_size, g_mf, g_emb, gEu, gEi, terms, w,
                                    n_terms, extra, c, total, f32_ticks, n_f32, u64_ticks, n_u64, workspace,
                                    workspace_bytes, mmssl_bpr_workspace_bytes(B), ticket, extra_parts, n_extra_parts);
  if (rc != 0) return rc;
  hipStream_t s = as_stream(stream);
  switch (d) {
    case 32: hipLaunchKernelGGL((bpr_step_kernel<8>), dim3(A.n_blocks), dim3(kBlock), 0, s, A); break;
    case 64: hipLaunchKernelGGL((bpr_step_kernel<16>), dim3(A.n_blocks), dim3(kBlock), 0, s, A); break;
    case 128: hipLaunchKernelGGL((bpr_step_kernel<32>), dim3(A.n_blocks), dim3(kBlock), 0, s, A); break;
    case 256: hipLaunchKernelGGL((bpr_step_kernel<64>), dim3(A.n_blocks), dim3(kBlock), 0, s, A); break;
  }
  MMSSL_LAUNCH_CHECK();
  return 0;
}
