// Step-level kernels that are not part of any one operator: the AdamW update of every trainable tensor
// in ONE launch, and the dropout keep-mask generator. Both keep their state (step counter / RNG
// counter) in device memory so that a captured hipGraph advances it on every replay.
//
// Reference: torch.optim.AdamW as constructed at /root/reference/MMSSL/main.py:76-80 (one param
// group, lr, default betas/eps/weight_decay=0.01, amsgrad=False) and nn.Dropout at Models.py:54.
#include <algorithm>

#include "common.hpp"

namespace mmssl {

// ---------------------------------------------------------------------------------------------
// AdamW
// ---------------------------------------------------------------------------------------------
struct AdamTensors {
  float* p[MMSSL_ADAMW_MAX_TENSORS];
  const float* g[MMSSL_ADAMW_MAX_TENSORS];
  float* m[MMSSL_ADAMW_MAX_TENSORS];
  float* v[MMSSL_ADAMW_MAX_TENSORS];
  int64_t n[MMSSL_ADAMW_MAX_TENSORS];
  int64_t gstride[MMSSL_ADAMW_MAX_TENSORS];           // floats between the slices of a sliced gradient
  int32_t slices[MMSSL_ADAMW_MAX_TENSORS];            // 1 = plain gradient; s > 1: g = g[0] + g[stride] + ... (in order)
  int32_t first_block[MMSSL_ADAMW_MAX_TENSORS + 1];   // prefix of per-tensor block counts
  int32_t count;
};

constexpr int kAdamPerThread = 4;                       // one float4 per array per thread
constexpr int kAdamPerBlock = kBlock * kAdamPerThread * 4;   // 4096 elements per block

// state[0] = number of completed steps (as float, like torch's capturable step tensor). It is advanced by
// a one-thread kernel launched right after this one: a "last block done" counter would serialise
// ~3.7 K same-address atomics at one L2 channel (measured: 68 us for the launch instead of ~15).
__global__ void tick_f32_kernel(float* __restrict__ counter) { counter[0] += 1.0f; }
__global__ void tick_u64_kernel(uint64_t* __restrict__ counter) { counter[0] += 1; }

// `pre_ticked`: state[0] already is the number of THIS step (a stream-ordered launch before this one advanced it:
// mmssl_step_tick / mmssl_loss_assemble_tick_f32); otherwise state[0] counts completed steps.
template <bool SLICED>
__device__ __forceinline__ void adamw_block(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                            float* __restrict__ v, int64_t n, int64_t base, int nsl, int64_t gst,
                                            float step_size, float bc2_sqrt, float decay, float beta1, float beta2,
                                            float eps) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t i = base + ((int64_t)r * kBlock + threadIdx.x) * kAdamPerThread;
    if (i + kAdamPerThread <= n) {
      float4 pp = *reinterpret_cast<float4*>(p + i);
      float4 gg = *reinterpret_cast<const float4*>(g + i);
      if (SLICED) {
        // split-K partial gradients, added in slice order; eight loads are requested before the first add (one memory
        // latency per eight slices instead of one per slice)
        for (int s0 = 1; s0 < nsl; s0 += 8) {
          float4 q[8];
#pragma unroll
          for (int k = 0; k < 8; ++k)
            q[k] = s0 + k < nsl ? *reinterpret_cast<const float4*>(g + (int64_t)(s0 + k) * gst + i)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int k = 0; k < 8; ++k)
            if (s0 + k < nsl) {
              gg.x += q[k].x; gg.y += q[k].y; gg.z += q[k].z; gg.w += q[k].w;
            }
        }
      }
      float4 mm = *reinterpret_cast<float4*>(m + i);
      float4 vv = *reinterpret_cast<float4*>(v + i);
      float* P = &pp.x;
      const float* G = &gg.x;
      float* M = &mm.x;
      float* V = &vv.x;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        P[c] *= decay;
        M[c] = M[c] + (G[c] - M[c]) * (1.0f - beta1);                 // exp_avg.lerp_(grad, 1 - beta1)
        V[c] = V[c] * beta2 + (1.0f - beta2) * G[c] * G[c];
        const float denom = sqrtf(V[c]) / bc2_sqrt + eps;
        P[c] -= step_size * (M[c] / denom);
      }
      *reinterpret_cast<float4*>(p + i) = pp;
      *reinterpret_cast<float4*>(m + i) = mm;
      *reinterpret_cast<float4*>(v + i) = vv;
    } else {
      for (int64_t j = i; j < n && j < i + kAdamPerThread; ++j) {
        float P = p[j] * decay;
        float G = g[j];
        if (SLICED)
          for (int sl = 1; sl < nsl; ++sl) G += g[(int64_t)sl * gst + j];
        const float M = m[j] + (G - m[j]) * (1.0f - beta1);
        const float V = v[j] * beta2 + (1.0f - beta2) * G * G;
        P -= step_size * (M / (sqrtf(V) / bc2_sqrt + eps));
        p[j] = P;
        m[j] = M;
        v[j] = V;
      }
    }
  }
}

__global__ __launch_bounds__(kBlock) void adamw_kernel(AdamTensors T, const float* __restrict__ state, float lr,
                                                       float beta1, float beta2, float eps, float wd, int pre_ticked) {
  int t = 0;
  while (t + 1 < T.count && (int)blockIdx.x >= T.first_block[t + 1]) ++t;
  const int64_t base = (int64_t)(blockIdx.x - T.first_block[t]) * kAdamPerBlock;
  // bias corrections once per block (powf is ~100s of instructions), broadcast through LDS;
  // same operation order as torch's _single_tensor_adamw / fused kernel in fp32
  __shared__ float sh[2];
  if (threadIdx.x == 0) {
    // pre-ticked with a counter nobody advanced would make the bias corrections divide by zero: treat it as step 1
    const float step = pre_ticked ? fmaxf(state[0], 1.0f) : state[0] + 1.0f;
    sh[0] = lr / (1.0f - powf(beta1, step));
    sh[1] = sqrtf(1.0f - powf(beta2, step));
  }
  __syncthreads();
  const float decay = 1.0f - lr * wd;
  // the sliced form (a block-uniform choice) lives in its own copy of the loop so that the plain one stays as it was
  if (T.slices[t] > 1)
    adamw_block<true>(T.p[t], T.g[t], T.m[t], T.v[t], T.n[t], base, T.slices[t], T.gstride[t], sh[0], sh[1], decay, beta1,
                      beta2, eps);
  else
    adamw_block<false>(T.p[t], T.g[t], T.m[t], T.v[t], T.n[t], base, 1, 0, sh[0], sh[1], decay, beta1, beta2, eps);
}

// ---------------------------------------------------------------------------------------------
// Dropout keep-mask: Philox4x32-10, one counter block (4 x 32 random bits) per 4 mask bytes.
// key = seed, counter = (element group, launch counter). keep[i] = (uniform[0,1) >= p).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t a = (uint64_t)0xD2511F53u * c[0];
  const uint64_t b = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(b >> 32) ^ c[1] ^ k0;
  const uint32_t n2 = (uint32_t)(a >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)b;
  c[3] = (uint32_t)a;
  c[0] = n0;
  c[2] = n2;
}

// state: uint64 seed; uint64 launch counter (advanced by tick_u64_kernel after the launch)
__global__ __launch_bounds__(kBlock) void dropout_mask_kernel(const uint64_t* __restrict__ state, float p, int64_t n4,
                                                              uint8_t* __restrict__ keep) {
  const uint64_t seed = state[0], launch = state[1];
  // threshold on the top 24 bits: keep iff u >= p with u = bits / 2^24 (exactly representable)
  const uint32_t thr = (uint32_t)(p * 16777216.0f);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) {
    uint32_t c[4] = {(uint32_t)i, (uint32_t)(i >> 32), (uint32_t)launch, (uint32_t)(launch >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      philox_round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    uchar4 o;
    o.x = (c[0] >> 8) >= thr;
    o.y = (c[1] >> 8) >= thr;
    o.z = (c[2] >> 8) >= thr;
    o.w = (c[3] >> 8) >= thr;
    reinterpret_cast<uchar4*>(keep)[i] = o;
  }
}

}  // namespace mmssl

using namespace mmssl;

extern "C" int mmssl_adamw_sliced_f32(float* const* params, const float* const* grads, float* const* exp_avg,
                                      float* const* exp_avg_sq, const int64_t* numel, const int32_t* slices,
                                      const int64_t* gstride, int count, float* state, float lr, float beta1,
                                      float beta2, float eps, float weight_decay, int external_tick, void* stream);

extern "C" int mmssl_adamw_ex_f32(float* const* params, const float* const* grads, float* const* exp_avg,
                                  float* const* exp_avg_sq, const int64_t* numel, int count, float* state, float lr,
                                  float beta1, float beta2, float eps, float weight_decay, int external_tick,
                                  void* stream) {
  return mmssl_adamw_sliced_f32(params, grads, exp_avg, exp_avg_sq, numel, nullptr, nullptr, count, state, lr, beta1,
                                beta2, eps, weight_decay, external_tick, stream);
}

extern "C" int mmssl_adamw_sliced_f32(float* const* params, const float* const* grads, float* const* exp_avg,
                                      float* const* exp_avg_sq, const int64_t* numel, const int32_t* slices,
                                      const int64_t* gstride, int count, float* state, float lr, float beta1,
                                      float beta2, float eps, float weight_decay, int external_tick, void* stream) {
  if (count < 0 || count > MMSSL_ADAMW_MAX_TENSORS || !state) return MMSSL_E_BADARG;
  if ((slices == nullptr) != (gstride == nullptr)) return MMSSL_E_BADARG;
  if (count == 0) return 0;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numel) return MMSSL_E_BADARG;
  AdamTensors T;
  int blocks = 0;
  for (int t = 0; t < count; ++t) {
    if (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t] || numel[t] <= 0) return MMSSL_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(params[t]) | reinterpret_cast<uintptr_t>(grads[t]) |
         reinterpret_cast<uintptr_t>(exp_avg[t]) | reinterpret_cast<uintptr_t>(exp_avg_sq[t])) & 15)
      return MMSSL_E_BADARG;
    T.p[t] = params[t];
    T.g[t] = grads[t];
    T.m[t] = exp_avg[t];
    T.v[t] = exp_avg_sq[t];
    T.n[t] = numel[t];
    T.slices[t] = slices ? slices[t] : 1;
    T.gstride[t] = slices ? gstride[t] : 0;
    if (T.slices[t] < 1 || (T.slices[t] > 1 && (T.gstride[t] < numel[t] || (T.gstride[t] & 3)))) return MMSSL_E_BADARG;
    T.first_block[t] = blocks;
    const int64_t nb = (numel[t] + kAdamPerBlock - 1) / kAdamPerBlock;
    if (nb > (1 << 24)) return MMSSL_E_UNSUPP;
    blocks += (int)nb;
  }
  T.first_block[count] = blocks;
  T.count = count;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, T,
                     (const float*)state, lr, beta1, beta2, eps, weight_decay, external_tick ? 1 : 0);
  MMSSL_LAUNCH_CHECK();
  if (!external_tick) {
    hipLaunchKernelGGL(tick_f32_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
    MMSSL_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int mmssl_adamw_f32(float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int64_t* numel, int count, float* state, float lr,
                               float beta1, float beta2, float eps, float weight_decay, void* stream) {
  return mmssl_adamw_ex_f32(params, grads, exp_avg, exp_avg_sq, numel, count, state, lr, beta1, beta2, eps,
                            weight_decay, 0, stream);
}

extern "C" int mmssl_dropout_mask_ex_u8(uint64_t* rng_state, float p, int64_t n, uint8_t* keep, int external_tick,
                                        void* stream) {
  if (!rng_state || !keep || n < 0 || (n & 3) || !(p >= 0.f && p < 1.f)) return MMSSL_E_BADARG;
  if ((reinterpret_cast<uintptr_t>(keep) & 3) || (reinterpret_cast<uintptr_t>(rng_state) & 7)) return MMSSL_E_BADARG;
  if (n == 0) return 0;
  const int64_t n4 = n / 4;
  const int blocks = (int)std::min<int64_t>((n4 + kBlock - 1) / kBlock, 2048);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream,
                     (const uint64_t*)rng_state, p, n4, keep);
  MMSSL_LAUNCH_CHECK();
  if (!external_tick) {
    hipLaunchKernelGGL(tick_u64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng_state + 1);
    MMSSL_LAUNCH_CHECK();
  }
  return 0;
}

namespace mmssl {
// dst[0 .. count) = ring[(step % n) * count ...]: a step picks its batch from a device-resident ring by a uint64 step
// counter of its own (the fp32 AdamW step counter stops incrementing at 2^24 steps and cannot index a ring for ever)
__global__ __launch_bounds__(kBlock) void select_slot_kernel(const int64_t* __restrict__ ring, int n, int64_t count,
                                                             const uint64_t* __restrict__ step_counter,
                                                             int64_t* __restrict__ dst) {
  const int64_t k = (int64_t)(step_counter[0] % (uint64_t)n);
  const int64_t* src = ring + k * count;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (int64_t)gridDim.x * kBlock) dst[i] = src[i];
}
}  // namespace mmssl

extern "C" int mmssl_select_slot_i64(const int64_t* ring, int n_slots, int64_t count, const uint64_t* step_counter,
                                     int64_t* dst, void* stream) {
  if (!ring || !dst || !step_counter || n_slots < 1 || count < 1) return MMSSL_E_BADARG;
  if (reinterpret_cast<uintptr_t>(step_counter) & 7) return MMSSL_E_BADARG;
  const int blocks = (int)std::min<int64_t>((count + kBlock - 1) / kBlock, 64);
  hipLaunchKernelGGL(select_slot_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, ring, n_slots, count,
                     step_counter, dst);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_tick_u64(uint64_t* counter, void* stream) {
  if (!counter || (reinterpret_cast<uintptr_t>(counter) & 7)) return MMSSL_E_BADARG;
  hipLaunchKernelGGL(tick_u64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_dropout_mask_u8(uint64_t* rng_state, float p, int64_t n, uint8_t* keep, void* stream) {
  return mmssl_dropout_mask_ex_u8(rng_state, p, n, keep, 0, stream);
}
