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
// Memory model: data stores, __threadfence_system(), then a system-scope release store of the flag; the waiter
// acquires at system scope before the consumer kernel (launched behind it in stream order) reads the window.
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

__global__ __launch_bounds__(kBlock) void peer_push_rows_kernel(const float* __restrict__ src, int64_t src_pitch,
                                                                int64_t rows, int w4, PeerPtrs dst, int64_t dst_row0,
                                                                int64_t dst_pitch, int world, int me, int ch, int max_ch,
                                                                PeerPtrs flags, uint32_t* __restrict__ epoch,
                                                                uint32_t* __restrict__ ticket) {
  const int64_t total = rows * w4;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / w4;
    const int c = (int)(i - r * w4);
    const float4 v = *reinterpret_cast<const float4*>(src + r * src_pitch + 4 * c);
    const int64_t o = (dst_row0 + r) * dst_pitch + 4 * c;
    // own link first, then the peers starting behind me: at any time the N ranks write to N different destinations
    for (int k = 0; k < world; ++k) {
      const int q = (me + k) % world;
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(dst.p[q]) + o) = v;
    }
  }
  __threadfence_system();
  __syncthreads();
  __shared__ int last;
  if (threadIdx.x == 0) last = (atomicAdd(ticket + ch, 1u) == gridDim.x - 1) ? 1 : 0;
  __syncthreads();
  if (!last) return;
  __shared__ uint32_t e;
  if (threadIdx.x == 0) {
    ticket[ch] = 0;
    e = epoch[ch] + 1;
    epoch[ch] = e;
  }
  __syncthreads();
  __threadfence_system();
  if ((int)threadIdx.x < world)
    __hip_atomic_store(reinterpret_cast<uint32_t*>(flags.p[threadIdx.x]) + (size_t)ch * kMaxWorld + me, e,
                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void peer_signal_kernel(int world, int me, int ch, PeerPtrs flags, uint32_t* __restrict__ epoch) {
  __shared__ uint32_t e;
  if (threadIdx.x == 0) {
    e = epoch[ch] + 1;
    epoch[ch] = e;
  }
  __syncthreads();
  __threadfence_system();
  if ((int)threadIdx.x < world)
    __hip_atomic_store(reinterpret_cast<uint32_t*>(flags.p[threadIdx.x]) + (size_t)ch * kMaxWorld + me, e,
                       __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void peer_wait_kernel(int world, int ch, const uint32_t* __restrict__ my_flags,
                                 const uint32_t* __restrict__ epoch, uint64_t timeout_ticks, uint32_t* __restrict__ err) {
  const int q = threadIdx.x;
  if (q >= world) return;
  const uint32_t want = epoch[ch];
  const uint32_t* f = my_flags + (size_t)ch * kMaxWorld + q;
  const uint64_t t0 = wall_clock64();
  // (int32_t)(have - want) >= 0: the epochs wrap after 2^32 calls on one channel
  while ((int32_t)(__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - want) < 0) {
    __builtin_amdgcn_s_sleep(64);
    if (wall_clock64() - t0 > timeout_ticks) {
      atomicOr(err, 1u << (q & 31));
      break;
    }
  }
  __threadfence_system();
}

__global__ __launch_bounds__(kBlock) void peer_pull_sum_kernel(PeerPtrs win, int world, int64_t row0, int64_t rows, int w4,
                                                               int64_t pitch, float* __restrict__ out, int64_t out_pitch) {
  const int64_t total = rows * w4;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t r = i / w4;
    const int c = (int)(i - r * w4);
    const int64_t o = (row0 + r) * pitch + 4 * c;
    float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(win.p[0]) + o);
    for (int q = 1; q < world; ++q) {
      const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(win.p[q]) + o);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    *reinterpret_cast<float4*>(out + r * out_pitch + 4 * c) = a;
  }
}

// out[j] = sum over slots q = 0 .. n - 1 (in that order) of slots[q * stride + j]: the local half of an all-reduce whose
// other half is every rank's push of its vector into slot [rank] of every window
__global__ __launch_bounds__(kBlock) void sum_slots_kernel(const float* __restrict__ slots, int n, int64_t stride,
                                                           int64_t len, float* __restrict__ out) {
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < len; j += (int64_t)gridDim.x * kBlock) {
    float a = slots[j];
    for (int q = 1; q < n; ++q) a += slots[(int64_t)q * stride + j];
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
                                        int width, int64_t dst_row0, int64_t dst_pitch, void* stream) {
  int rc = ready(p, ch);
  if (rc) return rc;
  if (!src || rows < 0 || width <= 0 || (width & 3) || (src_pitch & 3) || (dst_pitch & 3) || dst_row0 < 0 ||
      src_pitch < width || dst_pitch < width || ((uintptr_t)src & 15))
    return MMSSL_E_BADARG;
  PeerPtrs d;
  if ((rc = win_ptrs(p, win_id, &d)) != 0) return rc;
  if ((size_t)((dst_row0 + rows) * dst_pitch) * sizeof(float) > p->wins[win_id].bytes) return MMSSL_E_BADARG;
  hipLaunchKernelGGL(peer_push_rows_kernel, dim3(grid_for(rows * (width / 4))), dim3(kBlock), 0, as_stream(stream), src,
                     src_pitch, rows, width / 4, d, dst_row0, dst_pitch, p->world, p->rank, ch, p->max_channels,
                     flag_ptrs(p), p->epoch, p->ticket);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_peer_signal(mmssl_peer* p, int ch, void* stream) {
  int rc = ready(p, ch);
  if (rc) return rc;
  hipLaunchKernelGGL(peer_signal_kernel, dim3(1), dim3(64), 0, as_stream(stream), p->world, p->rank, ch, flag_ptrs(p),
                     p->epoch);
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
