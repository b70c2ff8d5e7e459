// Device-side exchange of row-sharded tables between the ranks of one node: IPC-mapped peer windows + epoch flags, no
// collective library in the data path (SURVEY.md section 5 "prefer the direct pattern": xGMI is point to point, every peer
// pushes its shard over its own link). The reference has no counterpart (MMSSL/main.py:529 picks ONE device); this is the
// transport under mmssl_amd/dist.py's exchanges, next to RCCL (kept as the A/B).
//
//   window   hipMalloc'd buffer of one rank that every peer has opened (hipIpcOpenMemHandle): a rank's GATHERED table
//            (peers push their rows into it) or its PARTIAL products (peers pull their rows out of it)
//   channel  one uint32 epoch per (channel, source rank) in every rank's flag block (fine-grained memory, opened by all
//            peers) + the rank's own epoch counter: `signal` = ++epoch, release-store it into slot [ch][me] of EVERY
//            rank's flags; `wait` = spin (bounded) until all slots [ch][*] of MY flags have reached my epoch.
//            All ranks run the same sequence of calls on a channel, so their epochs agree without any host handshake, and
//            the epoch lives in device memory: the kernels are hipGraph-capturable.
//   push     rows of a local table -> the same row range of every rank's window, then signal (last block to finish)
//   pull-sum out = sum over ranks q = 0 .. N-1, IN THAT ORDER, of rows [row0, row0 + rows) of rank q's window: the
//            reduce-scatter as a pull with a fixed summation order (bit-reproducible, identical on every rank)
//
// Memory model (round 6, second version). The first version fenced: every block of a push ran __threadfence_system()
// (buffer_wbl2 + buffer_inv: the XCD's whole L2 written back and dropped) and the waiter spun on ACQUIRE loads (an L2
// invalidate per poll) - 48 launches of ~28 us per step at world 1, 1.86 ms against 0.52 ms for the same step with identity
// exchanges, because the SpMMs running beside them live on their L2-resident tables. Now:
//   producer  rows are stored WRITE-THROUGH at system scope (relaxed system-scope atomic stores = `global_store ... sc0
//             sc1`: performed at the destination's memory, local HBM or a peer's over xGMI), `s_waitcnt vmcnt(0)` (gfx9
//             stores retire through vmcnt: the same instruction the compiler's release sequence ends with), an agent-scope
//             ticket; the block that takes the last ticket stores the epoch flags (system scope, fine-grained memory).
//             No cache-wide operation. Partials a plain kernel wrote into the local window (the SpMM's output) are made
//             visible by that kernel's END (its release writes the XCDs' L2s back); `signal` runs behind it in stream order
//             and adds one release fence of its own.
//   consumer  the wait kernel polls with RELAXED system-scope loads (they bypass the caches; no invalidate per poll); the
//             kernel that reads the window is a separate launch behind it, whose start invalidates the L2s' copies of
//             the window (local lines a peer has overwritten in HBM, or remote lines of a peer's window).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "peer.hip's visibility protocol is written for gfx942 / gfx950 memory semantics"
#endif
// Never hangs the device: a wait gives up after `timeout_ms`, sets the context's error word and returns; the host reads it
// with mmssl_peer_error (a wrong answer is then reported as an error, not consumed silently).
#include <cstring>
#include <vector>

#include "common.hpp"

using namespace mmssl;

namespace {
constexpr int kMaxWorld = 16;

struct Window {
  void* local = nullptr;
  size_t bytes = 0;
  void* peer[kMaxWorld] = {};      // [q] = rank q's window in this process's address space ([rank] = local)
  bool opened = false;
};

struct PeerPtrs {
  void* p[kMaxWorld];
};

// thread q < world of the calling block: spin (bounded) until rank q's epoch on channel `ch` has reached `want`
__device__ __forceinline__ void wait_channel(int world, int ch, const uint32_t* my_flags, uint32_t want,
                                             uint64_t timeout_ticks, uint32_t* err) {
  const int q = threadIdx.x;
  if (q >= world) return;
  const uint32_t* f = my_flags + (size_t)ch * kMaxWorld + q;
  const uint64_t t0 = wall_clock64();
  // once a wait of this context has given up, the later ones do not spin again: the step's results are void anyway, and
  // a job with dozens of waits per step must not stall a time-out's length at each of them before the host sees the error
  const bool failed = __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  // (int32_t)(have - want) >= 0: the epochs wrap after 2^32 calls on one channel
  while (!failed && (int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
    __builtin_amdgcn_s_sleep(64);
    if (wall_clock64() - t0 > timeout_ticks) {
      atomicOr(err, 1u << (q & 31));
      break;
    }
  }
}

// 16 bytes written through to the destination's memory (system scope) in ONE instruction. (Two 8-byte atomic stores - what
// the compiler offers - write half of every 32-byte sector per instruction: the push ran at 0.27 TB/s.) Inline asm is safe
// here: the data comes out of an ordinary load, whose wait the compiler inserts in front of the asm's operand read.
__device__ __forceinline__ void store16_sys(float* p, const float4& v) {
  typedef float floatx4 __attribute__((ext_vector_type(4)));
  const floatx4 d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(d) : "memory");
}

__global__ __launch_bounds__(kBlock) void peer_push_rows_kernel(const float* __restrict__ src, int64_t src_pitch,
                                                                int64_t rows, int w4, PeerPtrs dst, int64_t dst_row0,
                                                                int64_t dst_pitch, int world, int me, int ch, int max_ch,
                                                                PeerPtrs flags, uint32_t* __restrict__ epoch,
                                                                uint32_t* __restrict__ ticket, int wait_after,
                                                                const uint32_t* my_flags, uint64_t timeout_ticks,
                                                                uint32_t* err) {
  const int64_t total = rows * w4;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / w4;
    const int c = (int)(i - r * w4);
    const float4 v = *reinterpret_cast<const float4*>(src + r * src_pitch + 4 * c);
    const int64_t o = (dst_row0 + r) * dst_pitch + 4 * c;
    // own link first, then the peers starting behind me: at any time the N ranks write to N different destinations
    // (my own window: a plain cached store - only this device's later kernels read it; the write-through form measured
    // 0.3 TB/s on local memory, which a real job does not notice behind 7 links but a one-rank run does)
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(dst.p[me]) + o) = v;
    for (int k = 1; k < world; ++k) {
      const int q = (me + k) % world;
      store16_sys(reinterpret_cast<float*>(dst.p[q]) + o, v);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's rows are in their destinations' memory
  __syncthreads();
  __shared__ int last;
  if (threadIdx.x == 0)
    last = (__hip_atomic_fetch_add(ticket + ch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!last) return;
  __shared__ uint32_t e;
  if (threadIdx.x == 0) {
    __hip_atomic_store(ticket + ch, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    e = epoch[ch] + 1;
    epoch[ch] = e;
  }
  __syncthreads();
  if ((int)threadIdx.x < world)
    __hip_atomic_store(reinterpret_cast<uint32_t*>(flags.p[threadIdx.x]) + (size_t)ch * kMaxWorld + me, e,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // wait_after: this ONE block (the others have left) also waits for every peer's push - the gather is a single launch
  if (wait_after) {
    wait_channel(world, ch, my_flags, e, timeout_ticks, err);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
}

__global__ void peer_signal_kernel(int world, int me, int ch, PeerPtrs flags, uint32_t* __restrict__ epoch, int wait_after,
                                   const uint32_t* my_flags, uint64_t timeout_ticks, uint32_t* err) {
  __shared__ uint32_t e;
  if (threadIdx.x == 0) {
    e = epoch[ch] + 1;
    epoch[ch] = e;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");          // (one wave: the write-back half only, of the XCD it runs on)
  if ((int)threadIdx.x < world)
    __hip_atomic_store(reinterpret_cast<uint32_t*>(flags.p[threadIdx.x]) + (size_t)ch * kMaxWorld + me, e,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (wait_after) {                                      // signal + wait as ONE launch (the reduce-scatter, the step barrier)
    wait_channel(world, ch, my_flags, e, timeout_ticks, err);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
}

__global__ void peer_wait_kernel(int world, int ch, const uint32_t* __restrict__ my_flags,
                                 const uint32_t* __restrict__ epoch, uint64_t timeout_ticks, uint32_t* err) {
  wait_channel(world, ch, my_flags, epoch[ch], timeout_ticks, err);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");          // (the reader is the NEXT kernel: its start invalidates all L2s)
}

// 16 bytes read past the caches (system scope): the pull reads every element of the peers' windows exactly once, so it
// needs no cache - and then does not depend on its start having dropped stale copies of remote lines either.
__device__ __forceinline__ float4 load16_sys(const float* p) {
  typedef unsigned long long u64;
  u64* q = reinterpret_cast<u64*>(const_cast<float*>(p));
  const u64 lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const u64 hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  float4 r;
  __builtin_memcpy(&r.x, &lo, 8);
  __builtin_memcpy(&r.z, &hi, 8);
  return r;
}

// (The wait stays a launch of its own - ONE wave. A pull that waited inside itself was tried: its ~1000 spinning blocks fill
// the CUs, the kernels the peers are waiting for - this rank's pushes on other lanes, or another rank's work on a shared
// GPU - cannot be scheduled, and the waits time out: tests/test_dist_gpu.py world 3, two lanes.)
__global__ __launch_bounds__(kBlock) void peer_pull_sum_kernel(PeerPtrs win, int world, int64_t row0, int64_t rows, int w4,
                                                               int64_t pitch, float* __restrict__ out, int64_t out_pitch) {
  const int64_t total = rows * w4;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / w4;
    const int c = (int)(i - r * w4);
    const int64_t o = (row0 + r) * pitch + 4 * c;
    float4 a = load16_sys(reinterpret_cast<const float*>(win.p[0]) + o);
    for (int q = 1; q < world; ++q) {
      const float4 v = load16_sys(reinterpret_cast<const float*>(win.p[q]) + o);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(out + r * out_pitch + 4 * c) = a;
  }
}

// out[j] = sum over slots q = 0 .. n - 1 (in that order) of slots[q * stride + j]: the local half of an all-reduce whose
// other half is every rank's push of its vector into slot [rank] of every window
__global__ __launch_bounds__(kBlock) void sum_slots_kernel(const float* slots, int n, int64_t stride, int64_t len,
                                                           float* __restrict__ out) {
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < len; j += (int64_t)gridDim.x * kBlock) {
    float a = __hip_atomic_load(const_cast<float*>(slots) + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    for (int q = 1; q < n; ++q)
      a += __hip_atomic_load(const_cast<float*>(slots) + (int64_t)q * stride + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    out[j] = a;
  }
}

inline unsigned grid_for(int64_t total) {
  int64_t b = (total + kBlock - 1) / kBlock;
  return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
}  // namespace

struct mmssl_peer {
  int world = 1, rank = 0, max_channels = 0;
  uint32_t* flags_local = nullptr;          // [max_channels][kMaxWorld]
  bool flags_finegrained = false;
  void* flags_peer[kMaxWorld] = {};
  bool flags_opened = false;
  uint32_t* epoch = nullptr;                // [max_channels]
  uint32_t* ticket = nullptr;               // [max_channels]
  uint32_t* err = nullptr;
  uint64_t timeout_ticks = 20ull * 100000000ull;      // wall_clock64 counts 100 MHz
  std::vector<Window> wins;
};

extern "C" int mmssl_peer_create(int world, int rank, int max_channels, mmssl_peer** out) {
  if (!out || world < 1 || world > kMaxWorld || rank < 0 || rank >= world || max_channels < 1) return MMSSL_E_BADARG;
  auto* p = new mmssl_peer();
  p->world = world;
  p->rank = rank;
  p->max_channels = max_channels;
  const size_t fb = (size_t)max_channels * kMaxWorld * sizeof(uint32_t);
  // flags: fine-grained (coherent across devices without cache maintenance) when the runtime exports it over IPC
  hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void**>(&p->flags_local), fb, hipDeviceMallocFinegrained);
  p->flags_finegrained = (e == hipSuccess);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipMalloc(reinterpret_cast<void**>(&p->flags_local), fb);
  }
  if (e != hipSuccess) { delete p; return (int)e; }
  const size_t cb = (size_t)max_channels * sizeof(uint32_t);
  if ((e = hipMalloc(reinterpret_cast<void**>(&p->epoch), cb)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&p->ticket), cb)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&p->err), sizeof(uint32_t))) != hipSuccess) {
    delete p;
    return (int)e;
  }
  MMSSL_HIP_TRY(hipMemset(p->flags_local, 0, fb));
  MMSSL_HIP_TRY(hipMemset(p->epoch, 0, cb));
  MMSSL_HIP_TRY(hipMemset(p->ticket, 0, cb));
  MMSSL_HIP_TRY(hipMemset(p->err, 0, sizeof(uint32_t)));
  MMSSL_HIP_TRY(hipDeviceSynchronize());
  p->flags_peer[rank] = p->flags_local;
  *out = p;
  return 0;
}

extern "C" int mmssl_peer_destroy(mmssl_peer* p) {
  if (!p) return 0;
  (void)hipDeviceSynchronize();
  for (auto& w : p->wins) {
    for (int q = 0; q < p->world; ++q)
      if (q != p->rank && w.peer[q]) (void)hipIpcCloseMemHandle(w.peer[q]);
    if (w.local) (void)hipFree(w.local);
  }
  for (int q = 0; q < p->world; ++q)
    if (q != p->rank && p->flags_peer[q]) (void)hipIpcCloseMemHandle(p->flags_peer[q]);
  if (p->flags_local) (void)hipFree(p->flags_local);
  if (p->epoch) (void)hipFree(p->epoch);
  if (p->ticket) (void)hipFree(p->ticket);
  if (p->err) (void)hipFree(p->err);
  delete p;
  return 0;
}

extern "C" int mmssl_peer_info(const mmssl_peer* p, int64_t* info) {
  if (!p || !info) return MMSSL_E_BADARG;
  info[0] = p->world;
  info[1] = p->rank;
  info[2] = p->max_channels;
  info[3] = p->flags_finegrained ? 1 : 0;
  info[4] = (int64_t)p->wins.size();
  size_t b = 0;
  for (const auto& w : p->wins) b += w.bytes;
  info[5] = (int64_t)b;
  return 0;
}

extern "C" int mmssl_peer_set_timeout_ms(mmssl_peer* p, int64_t ms) {
  if (!p || ms < 1) return MMSSL_E_BADARG;
  p->timeout_ticks = (uint64_t)ms * 100000ull;
  return 0;
}

extern "C" int mmssl_peer_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

extern "C" int mmssl_peer_flags_handle(mmssl_peer* p, void* handle_out) {
  if (!p || !handle_out) return MMSSL_E_BADARG;
  hipIpcMemHandle_t h;
  MMSSL_HIP_TRY(hipIpcGetMemHandle(&h, p->flags_local));
  memcpy(handle_out, &h, sizeof(h));
  return 0;
}

extern "C" int mmssl_peer_open_flags(mmssl_peer* p, const void* handles) {
  if (!p || !handles || p->flags_opened) return MMSSL_E_BADARG;
  for (int q = 0; q < p->world; ++q) {
    if (q == p->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, reinterpret_cast<const char*>(handles) + (size_t)q * sizeof(h), sizeof(h));
    MMSSL_HIP_TRY(hipIpcOpenMemHandle(&p->flags_peer[q], h, hipIpcMemLazyEnablePeerAccess));
  }
  p->flags_opened = true;
  return 0;
}

extern "C" int mmssl_peer_window_create(mmssl_peer* p, int64_t bytes, int* win_id, void* handle_out, void** local_ptr) {
  if (!p || bytes <= 0 || !win_id || !handle_out || !local_ptr) return MMSSL_E_BADARG;
  Window w;
  w.bytes = (size_t)((bytes + 255) / 256 * 256);
  MMSSL_HIP_TRY(hipMalloc(&w.local, w.bytes));
  MMSSL_HIP_TRY(hipMemset(w.local, 0, w.bytes));
  MMSSL_HIP_TRY(hipDeviceSynchronize());
  w.peer[p->rank] = w.local;
  hipIpcMemHandle_t h;
  MMSSL_HIP_TRY(hipIpcGetMemHandle(&h, w.local));
  memcpy(handle_out, &h, sizeof(h));
  *win_id = (int)p->wins.size();
  *local_ptr = w.local;
  p->wins.push_back(w);
  return 0;
}

extern "C" int mmssl_peer_window_open(mmssl_peer* p, int win_id, const void* handles) {
  if (!p || !handles || win_id < 0 || win_id >= (int)p->wins.size() || p->wins[win_id].opened) return MMSSL_E_BADARG;
  Window& w = p->wins[win_id];
  for (int q = 0; q < p->world; ++q) {
    if (q == p->rank) continue;
    hipIpcMemHandle_t h;
    memcpy(&h, reinterpret_cast<const char*>(handles) + (size_t)q * sizeof(h), sizeof(h));
    MMSSL_HIP_TRY(hipIpcOpenMemHandle(&w.peer[q], h, hipIpcMemLazyEnablePeerAccess));
  }
  w.opened = true;
  return 0;
}

namespace {
inline int ready(const mmssl_peer* p, int ch) {
  if (!p || ch < 0 || ch >= p->max_channels) return MMSSL_E_BADARG;
  if (p->world > 1 && !p->flags_opened) return MMSSL_E_BADARG;
  return 0;
}
inline PeerPtrs flag_ptrs(const mmssl_peer* p) {
  PeerPtrs f;
  for (int q = 0; q < kMaxWorld; ++q) f.p[q] = q < p->world ? p->flags_peer[q] : nullptr;
  return f;
}
inline int win_ptrs(const mmssl_peer* p, int win_id, PeerPtrs* out) {
  if (win_id < 0 || win_id >= (int)p->wins.size()) return MMSSL_E_BADARG;
  const Window& w = p->wins[win_id];
  if (p->world > 1 && !w.opened) return MMSSL_E_BADARG;
  for (int q = 0; q < kMaxWorld; ++q) out->p[q] = q < p->world ? w.peer[q] : nullptr;
  return 0;
}
}  // namespace

extern "C" int mmssl_peer_push_rows_f32(mmssl_peer* p, int ch, int win_id, const float* src, int64_t src_pitch, int64_t rows,
                                        int width, int64_t dst_row0, int64_t dst_pitch, int wait_after, void* stream) {
  int rc = ready(p, ch);
  if (rc) return rc;
  if (!src || rows < 0 || width <= 0 || (width & 3) || (src_pitch & 3) || (dst_pitch & 3) || dst_row0 < 0 ||
      src_pitch < width || dst_pitch < width || ((uintptr_t)src & 15))
    return MMSSL_E_BADARG;
  PeerPtrs d;
  if ((rc = win_ptrs(p, win_id, &d)) != 0) return rc;
  if ((size_t)((dst_row0 + rows) * dst_pitch) * sizeof(float) > p->wins[win_id].bytes) return MMSSL_E_BADARG;
  // at most one block per CU: every block ends with an agent-scope ticket on ONE address, and those serialise at the memory
  // side (~15 ns each across 8 XCDs: with one block per 256 elements a 4.7 MB push spent 17 us, most of it in 1147 tickets)
  unsigned grid = grid_for(rows * (width / 4));
  if (grid > 256u) grid = 256u;
  hipLaunchKernelGGL(peer_push_rows_kernel, dim3(grid), dim3(kBlock), 0, as_stream(stream), src,
                     src_pitch, rows, width / 4, d, dst_row0, dst_pitch, p->world, p->rank, ch, p->max_channels,
                     flag_ptrs(p), p->epoch, p->ticket, wait_after ? 1 : 0, p->flags_local, p->timeout_ticks, p->err);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_peer_signal(mmssl_peer* p, int ch, void* stream) {
  int rc = ready(p, ch);
  if (rc) return rc;
  hipLaunchKernelGGL(peer_signal_kernel, dim3(1), dim3(64), 0, as_stream(stream), p->world, p->rank, ch, flag_ptrs(p),
                     p->epoch, 0, p->flags_local, p->timeout_ticks, p->err);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_peer_signal_wait(mmssl_peer* p, int ch, void* stream) {
  int rc = ready(p, ch);
  if (rc) return rc;
  hipLaunchKernelGGL(peer_signal_kernel, dim3(1), dim3(64), 0, as_stream(stream), p->world, p->rank, ch, flag_ptrs(p),
                     p->epoch, 1, p->flags_local, p->timeout_ticks, p->err);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_peer_wait(mmssl_peer* p, int ch, void* stream) {
  int rc = ready(p, ch);
  if (rc) return rc;
  hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(64), 0, as_stream(stream), p->world, ch, p->flags_local, p->epoch,
                     p->timeout_ticks, p->err);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_peer_pull_sum_rows_f32(mmssl_peer* p, int win_id, int64_t row0, int64_t rows, int width, int64_t pitch,
                                            float* out, int64_t out_pitch, void* stream) {
  if (!p || !out || rows < 0 || width <= 0 || (width & 3) || (pitch & 3) || (out_pitch & 3) || row0 < 0 || pitch < width ||
      out_pitch < width || ((uintptr_t)out & 15))
    return MMSSL_E_BADARG;
  PeerPtrs w;
  int rc = win_ptrs(p, win_id, &w);
  if (rc) return rc;
  if ((size_t)((row0 + rows) * pitch) * sizeof(float) > p->wins[win_id].bytes) return MMSSL_E_BADARG;
  hipLaunchKernelGGL(peer_pull_sum_kernel, dim3(grid_for(rows * (width / 4))), dim3(kBlock), 0, as_stream(stream), w,
                     p->world, row0, rows, width / 4, pitch, out, out_pitch);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_peer_sum_slots_f32(const float* slots, int n, int64_t stride, int64_t len, float* out, void* stream) {
  if (!slots || !out || n < 1 || len < 0 || stride < len) return MMSSL_E_BADARG;
  if (len == 0) return 0;
  hipLaunchKernelGGL(sum_slots_kernel, dim3(grid_for(len)), dim3(kBlock), 0, as_stream(stream), slots, n, stride, len, out);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_peer_error(mmssl_peer* p, uint32_t* err) {
  if (!p || !err) return MMSSL_E_BADARG;
  MMSSL_HIP_TRY(hipMemcpy(err, p->err, sizeof(uint32_t), hipMemcpyDeviceToHost));
  return 0;
}
