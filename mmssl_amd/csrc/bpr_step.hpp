// The hot step's loss tail (BPR backward + BPR loss + loss assembly + counter ticks) as a DEVICE function, so that it can
// be a kernel of its own (bpr.hip: bpr_step_kernel) or ride along as extra blocks of another launch of the loss chain
// (infonce.hip: the InfoNCE backward pair-tile kernel, which nothing in the BPR part depends on).
// Reference: /root/reference/MMSSL/main.py:368-371, 420, 499-511. See mmssl_bpr_step_f32 in include/mmssl_hip.h.
#pragma once
#include "common.hpp"

namespace mmssl {

struct StepTicks {
  float* f32[4];
  unsigned long long* u64[4];
  int n_f32, n_u64;
};

struct BprStepArgs {
  const float4* Eu;
  const float4* Ei;
  const int64_t* users;
  const int64_t* pos;
  const int64_t* neg;
  int64_t B;
  float decay;
  int64_t batch_size;
  const float* g_mf;
  const float* g_emb;
  float* gEu;
  float* gEi;
  float* part;
  int* ticket;
  float* terms;
  const float* w;
  int n_terms;
  const float* extra;
  float cex;
  float* total;
  StepTicks T;
  const float* xparts;
  int64_t n_xparts;
  float* extra_out;
  int n_blocks;            // BPR blocks: (B + kBlock / LPR - 1) / (kBlock / LPR)
};

__device__ __forceinline__ float bpr_log_sigmoid(float x) {
  // min(x,0) - log1p(exp(-|x|))  (what F.logsigmoid computes)
  return fminf(x, 0.f) - log1pf(expf(-fabsf(x)));
}

// ---- the tail in two parts ------------------------------------------------------------------------------------------
// rows part: one BPR block, `block` in [0, A.n_blocks): gathers, scores, the scatter-added gradients and this block's two
// partial sums (log-sigmoid, squared norms). Returns them (valid in thread 0).
template <int LPR>
__device__ __forceinline__ void bpr_rows_part(const BprStepArgs& A, int block, float* red, float& t0, float& t1) {
  constexpr int GPB = kBlock / LPR;
  constexpr int D = LPR * 4;
  const int lig = threadIdx.x & (LPR - 1);
  const int64_t b = (int64_t)block * GPB + threadIdx.x / LPR;
  float ls = 0.f, sq = 0.f;
  if (b < A.B) {
    const int64_t ru = A.users[b], rp = A.pos[b], rn = A.neg[b];
    const float4 u = A.Eu[ru * LPR + lig], p = A.Ei[rp * LPR + lig], n = A.Ei[rn * LPR + lig];
    const float sp = group_sum<LPR>(f4_dot(u, p));
    const float sn = group_sum<LPR>(f4_dot(u, n));
    const float q = group_sum<LPR>(f4_dot(u, u) + f4_dot(p, p) + f4_dot(n, n));
    const float diff = sp - sn;
    if (lig == 0) {
      ls = bpr_log_sigmoid(diff);
      sq = q;
    }
    const float sig_neg = 1.f / (1.f + expf(diff));
    const float cm = -A.g_mf[0] * sig_neg / (float)A.B;
    const float ce = A.g_emb[0] * A.decay / (float)A.batch_size;
    float* du = A.gEu + ru * D + lig * 4;
    float* dp = A.gEi + rp * D + lig * 4;
    float* dn = A.gEi + rn * D + lig * 4;
    unsafeAtomicAdd(du + 0, cm * (p.x - n.x) + ce * u.x); unsafeAtomicAdd(du + 1, cm * (p.y - n.y) + ce * u.y);
    unsafeAtomicAdd(du + 2, cm * (p.z - n.z) + ce * u.z); unsafeAtomicAdd(du + 3, cm * (p.w - n.w) + ce * u.w);
    unsafeAtomicAdd(dp + 0, cm * u.x + ce * p.x); unsafeAtomicAdd(dp + 1, cm * u.y + ce * p.y);
    unsafeAtomicAdd(dp + 2, cm * u.z + ce * p.z); unsafeAtomicAdd(dp + 3, cm * u.w + ce * p.w);
    unsafeAtomicAdd(dn + 0, -cm * u.x + ce * n.x); unsafeAtomicAdd(dn + 1, -cm * u.y + ce * n.y);
    unsafeAtomicAdd(dn + 2, -cm * u.z + ce * n.z); unsafeAtomicAdd(dn + 3, -cm * u.w + ce * n.w);
  }
  t0 = block_sum_256(ls, red);
  t1 = block_sum_256(sq, red);
}

// assembly part (ONE block, every block's partials visible): reduces the partials in block order (the arithmetic of
// bpr_finalize_kernel), the extra term's partials if given, writes terms[0..2], assembles total, advances the counters.
// `xsum_ready`: the extra term's sum already sits behind the partials (the one-launch form's block 0 put it there).
__device__ __forceinline__ void bpr_assemble_part(const BprStepArgs& A, float* red, bool atomic_loads, bool xsum_ready) {
  const int nblocks = A.n_blocks;
  const unsigned* part = reinterpret_cast<const unsigned*>(A.part);
  auto ld = [&](int i) {
    return atomic_loads ? __uint_as_float(__hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        : A.part[i];
  };
  float a = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < nblocks; i += kBlock) {
    a += ld(2 * i + 0);
    c += ld(2 * i + 1);
  }
  const float lsum = block_sum_256(a, red);
  const float qsum = block_sum_256(c, red);
  float xs = 0.f;
  if (A.xparts) {
    if (xsum_ready) {
      xs = ld(2 * nblocks);
    } else {                        // the arithmetic of sum_partials_kernel
      float e = 0.f;
      for (int64_t i = threadIdx.x; i < A.n_xparts; i += kBlock) e += A.xparts[i];
      xs = block_sum_256(e, red);
    }
    if (threadIdx.x == 0 && A.extra_out) A.extra_out[0] = xs;
  } else if (A.extra) {
    xs = A.extra[0];
  }
  if (threadIdx.x == 0) {
    const float t_mf = -(lsum / (float)A.B), t_emb = A.decay * ((0.5f * qsum) / (float)A.batch_size);
    A.terms[0] = t_mf;
    A.terms[1] = t_emb;
    A.terms[2] = 0.f;
    float t = A.w[0] * t_mf + A.w[1] * t_emb + A.w[2] * 0.f;          // same order as loss_assemble_kernel
    for (int k = 3; k < A.n_terms; ++k) t += A.w[k] * A.terms[k];
    if (A.extra || A.xparts) t += A.cex * xs;
    A.total[0] = t;
    for (int k = 0; k < A.T.n_f32; ++k) A.T.f32[k][0] += 1.0f;
    for (int k = 0; k < A.T.n_u64; ++k) A.T.u64[k][0] += 1ull;
  }
}

// One BPR block of the ONE-launch tail; `block` in [0, A.n_blocks). The LAST block to arrive runs the assembly part.
template <int LPR>
__device__ __forceinline__ void bpr_step_block(const BprStepArgs& A, int block) {
  __shared__ float red[4];
  __shared__ int s_last;
  const int nblocks = A.n_blocks;
  float t0, t1;
  bpr_rows_part<LPR>(A, block, red, t0, t1);
  unsigned* part = reinterpret_cast<unsigned*>(A.part);
  // the extra term given as partial sums (the forward's regulariser partials): block 0 reduces them (the arithmetic of
  // sum_partials_kernel) next to the other blocks' work and parks the value behind the BPR partials
  if (A.xparts && block == 0) {
    float e = 0.f;
    for (int64_t i = threadIdx.x; i < A.n_xparts; i += kBlock) e += A.xparts[i];
    const float xsum = block_sum_256(e, red);
    if (threadIdx.x == 0)
      __hip_atomic_store(part + 2 * nblocks, __float_as_uint(xsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(part + 2 * block + 0, __float_as_uint(t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(part + 2 * block + 1, __float_as_uint(t1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    const int prev = __hip_atomic_fetch_add(A.ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (prev == nblocks - 1);
    if (s_last) __hip_atomic_store(A.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // re-arm
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  bpr_assemble_part(A, red, true, true);
}

// The TWO-launch form (the hot step's loss chain): the rows part rides as guest blocks of an EARLY launch of the chain
// (the InfoNCE prep: it depends on nothing there), the assembly part as one guest block of the chain's LAST launch.
template <int LPR>
__device__ __forceinline__ void bpr_rows_block(const BprStepArgs& A, int block) {
  __shared__ float red[4];
  float t0, t1;
  bpr_rows_part<LPR>(A, block, red, t0, t1);
  if (threadIdx.x == 0) {
    A.part[2 * block + 0] = t0;
    A.part[2 * block + 1] = t1;
  }
}
__device__ __forceinline__ void bpr_assemble_block(const BprStepArgs& A) {
  __shared__ float red[4];
  bpr_assemble_part(A, red, false, false);
}

// Validates the C-ABI arguments of mmssl_bpr_step_f32 and fills the kernel argument block (host side).
inline int make_bpr_step_args(BprStepArgs& A, const float* Eu, const float* Ei, const int64_t* users, const int64_t* pos,
                              const int64_t* neg, int64_t B, int d, float decay, int64_t batch_size, const float* g_mf,
                              const float* g_emb, float* gEu, float* gEi, float* terms, const float* w, int n_terms,
                              const float* extra, float c, float* total, float* const* f32_ticks, int n_f32,
                              uint64_t* const* u64_ticks, int n_u64, void* workspace, size_t workspace_bytes,
                              size_t need_bytes, int* ticket, const float* extra_parts, int64_t n_extra_parts) {
  if (B <= 0 || batch_size <= 0 || !Eu || !Ei || !users || !pos || !neg || !g_mf || !g_emb || !gEu || !gEi)
    return MMSSL_E_BADARG;
  if (!terms || !w || !total || !ticket || n_terms < 3 || n_terms > 16) return MMSSL_E_BADARG;
  if (extra_parts && (n_extra_parts <= 0 || !extra)) return MMSSL_E_BADARG;
  if (n_f32 < 0 || n_f32 > 4 || n_u64 < 0 || n_u64 > 4 || (n_f32 > 0 && !f32_ticks) || (n_u64 > 0 && !u64_ticks))
    return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  if (!workspace || workspace_bytes < need_bytes) return MMSSL_E_WORKSPACE;
  A.Eu = reinterpret_cast<const float4*>(Eu);
  A.Ei = reinterpret_cast<const float4*>(Ei);
  A.users = users; A.pos = pos; A.neg = neg;
  A.B = B; A.decay = decay; A.batch_size = batch_size;
  A.g_mf = g_mf; A.g_emb = g_emb; A.gEu = gEu; A.gEi = gEi;
  A.part = reinterpret_cast<float*>(workspace);
  A.ticket = ticket;
  A.terms = terms; A.w = w; A.n_terms = n_terms;
  A.extra = extra_parts ? nullptr : extra;
  A.cex = c;
  A.total = total;
  A.T.n_f32 = n_f32; A.T.n_u64 = n_u64;
  for (int k = 0; k < 4; ++k) {
    A.T.f32[k] = k < n_f32 ? f32_ticks[k] : nullptr;
    A.T.u64[k] = k < n_u64 ? reinterpret_cast<unsigned long long*>(u64_ticks[k]) : nullptr;
    if ((k < n_f32 && !A.T.f32[k]) || (k < n_u64 && !A.T.u64[k])) return MMSSL_E_BADARG;
  }
  A.xparts = extra_parts;
  A.n_xparts = n_extra_parts;
  A.extra_out = extra_parts ? const_cast<float*>(extra) : nullptr;       // the reduced value is stored there
  const int gpb = kBlock / (d / 4);
  A.n_blocks = (int)((B + gpb - 1) / gpb);
  return 0;
}

}  // namespace mmssl
