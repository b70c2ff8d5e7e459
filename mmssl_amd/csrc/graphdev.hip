// Graph plans built ON THE DEVICE from (user, item) pairs: the reference's modal-graph rebuild
// (/root/reference/MMSSL/main.py:378-405: csr_matrix(ones, (x, y)) -> csr_norm(mean_flag=True) of the matrix and of
// its transpose -> two torch sparse tensors, every T batches) without the python-list / scipy / host-plan / upload
// round trip. One pair list yields BOTH plans of the rebuild (A_ui row-normalised by user degree, A_iu by item
// degree), each with its transposed direction for the backward.
//
//   * duplicates stay separate edges: a pair that occurs c times contributes c edges of weight 1/sqrt(deg(row)),
//     which sums to the reference's c / sqrt(rowsum + 1e-8) (csr_matrix sums duplicates, main.py:379-380);
//   * pairs are ordered by one single-block bitonic sort in LDS (<= 16384 pairs: the rebuild handles
//     batch_size * k * T pairs), row pointers by binary search, work lists by the rules of the host planner
//     (graph.hip) — compacted with atomics, so the ORDER of work items varies from build to build; every row's own
//     summation order is fixed by the sorted pairs, i.e. results are bitwise reproducible;
//   * item counts stay in device memory (DirPlan::dyn) and SpMM launches use capacity-sized grids, so a rebuild
//     never synchronises with the host and can be replayed inside a hipGraph; all buffers are allocated once.
#include <algorithm>
#include <new>

#include "graph_internal.hpp"

using namespace mmssl;

namespace {

constexpr int kMaxPairs = 16384;           // 128 KB of 64-bit keys in LDS
constexpr int kSortThreads = 1024;
constexpr int kShort = 32, kTask = 128;    // the host planner's defaults (MMSSL_PLAN_SHORT_MAX / TASK_NNZ)

// keys[e] = (major << 32) | minor, sorted ascending; padding sorts last
__global__ __launch_bounds__(kSortThreads) void pairs_sort_kernel(const int64_t* __restrict__ major,
                                                                  const int64_t* __restrict__ minor, int n,
                                                                  unsigned long long* __restrict__ out) {
  extern __shared__ unsigned long long keys[];
  int m = 1;
  while (m < n) m <<= 1;
  for (int i = threadIdx.x; i < m; i += kSortThreads)
    keys[i] = i < n ? (((unsigned long long)major[i] << 32) | (unsigned long long)(unsigned)minor[i]) : ~0ull;
  __syncthreads();
  for (int size = 2; size <= m; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < (m >> 1); i += kSortThreads) {
        const int lo = 2 * i - (i & (stride - 1));           // index with bit `stride` clear
        const int hi = lo + stride;
        const bool asc = (lo & size) == 0;
        const unsigned long long a = keys[lo], b = keys[hi];
        if ((a > b) == asc) { keys[lo] = b; keys[hi] = a; }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < n; i += kSortThreads) out[i] = keys[i];
}

__global__ __launch_bounds__(kBlock) void rowptr_kernel(const unsigned long long* __restrict__ keys, int n, int rows,
                                                        int32_t* __restrict__ rowptr) {
  const int r = blockIdx.x * kBlock + threadIdx.x;
  if (r > rows) return;
  const unsigned long long t = (unsigned long long)r << 32;       // first key of row r
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < t) lo = mid + 1;
    else hi = mid;
  }
  rowptr[r] = lo;
}

// edges of the major-sorted list: {minor, w} with w = 1/sqrt(deg of the NORMALISING side); two weightings at once
__global__ __launch_bounds__(kBlock) void edges_kernel(const unsigned long long* __restrict__ keys, int n,
                                                       const int32_t* __restrict__ rp_major,
                                                       const int32_t* __restrict__ rp_minor, Edge* __restrict__ by_major,
                                                       Edge* __restrict__ by_minor) {
  const int e = blockIdx.x * kBlock + threadIdx.x;
  if (e >= n) return;
  const unsigned long long k = keys[e];
  const int maj = (int)(k >> 32), mnr = (int)(k & 0xffffffffu);
  const float dmaj = (float)(rp_major[maj + 1] - rp_major[maj]);
  const float dmin = (float)(rp_minor[mnr + 1] - rp_minor[mnr]);
  by_major[e].col = mnr;
  by_major[e].val = 1.0f / sqrtf(dmaj + 1e-8f);       // rows of this grouping normalise (A_ui forward / A_iu forward)
  by_minor[e].col = mnr;
  by_minor[e].val = 1.0f / sqrtf(dmin + 1e-8f);       // the other plan's transposed direction: its rows are `minor`
}

// counts: [0] n_g  [1] n_w (heavy items, then + light)  [2] n_multi  [3] n_slots  [4] nnz  [5] heavy items  [6] light
__global__ __launch_bounds__(kBlock) void plan_pass1_kernel(const int32_t* __restrict__ rowptr, int rows,
                                                            int4* __restrict__ gitems, int4* __restrict__ witems,
                                                            int4* __restrict__ multi, int32_t* __restrict__ slot2multi,
                                                            int32_t* __restrict__ cnt) {
  const int r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  const int beg = rowptr[r], end = rowptr[r + 1], deg = end - beg;
  if (deg <= kShort) {
    gitems[atomicAdd(&cnt[0], 1)] = make_int4(r, beg, end, -1);
  } else if (deg > kTask) {
    const int t = (deg + kTask - 1) / kTask, nb = (t + 3) / 4;
    int slot = 0;
    if (nb > 1) {
      slot = atomicAdd(&cnt[3], nb);
      const int m = atomicAdd(&cnt[2], 1);
      multi[m] = make_int4(r, slot, nb, 0);
      for (int j = 0; j < nb; ++j) slot2multi[slot + j] = m;
    }
    const int base = atomicAdd(&cnt[5], nb * 4);
    for (int k = 0; k < nb * 4; ++k) {
      const int code = nb > 1 ? slot + k / 4 : -2;
      witems[base + k] = k < t ? make_int4(r, beg + k * kTask, min(end, beg + (k + 1) * kTask), code)
                               : make_int4(-1, 0, 0, code);
    }
  }
}
__global__ __launch_bounds__(kBlock) void plan_pass2_kernel(const int32_t* __restrict__ rowptr, int rows,
                                                            int4* __restrict__ witems, int32_t* __restrict__ cnt) {
  const int r = blockIdx.x * kBlock + threadIdx.x;
  if (r >= rows) return;
  const int beg = rowptr[r], end = rowptr[r + 1], deg = end - beg;
  if (deg > kShort && deg <= kTask) witems[cnt[5] + atomicAdd(&cnt[6], 1)] = make_int4(r, beg, end, -1);
}
__global__ void plan_finish_kernel(int32_t* __restrict__ cnt, int n) {
  cnt[1] = cnt[5] + cnt[6];
  cnt[4] = n;
}
__global__ void zero_counts_kernel(int32_t* a, int32_t* b, int32_t* c, int32_t* d) {
  if (threadIdx.x < 8) a[threadIdx.x] = b[threadIdx.x] = c[threadIdx.x] = d[threadIdx.x] = 0;
}

template <typename T>
int dev_alloc(T** p, size_t n) {
  *p = nullptr;
  MMSSL_HIP_TRY(hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(T)));
  return 0;
}

int alloc_dir(DirPlan& p, int32_t rows, int32_t cols, int cap) {
  p.rows = rows;
  p.cols = cols;
  p.nnz = cap;
  p.n_g = rows;                                              // capacities (see DirPlan::dyn)
  p.n_w = cap / (kShort + 1) + cap / kTask + 4 * (cap / (kTask + 1)) + 8;
  p.n_multi = cap / (4 * kTask + 1) + 1;
  p.n_slots = cap / (4 * kTask) + p.n_multi + 1;
  int rc = 0;
  if ((rc = dev_alloc(&p.rowptr, (size_t)rows + 1))) return rc;
  if ((rc = dev_alloc(&p.edges, (size_t)cap))) return rc;
  if ((rc = dev_alloc(&p.gitems, (size_t)p.n_g))) return rc;
  if ((rc = dev_alloc(&p.witems, (size_t)p.n_w))) return rc;
  if ((rc = dev_alloc(&p.multi, (size_t)p.n_multi))) return rc;
  if ((rc = dev_alloc(&p.slot2multi, (size_t)p.n_slots))) return rc;
  if ((rc = dev_alloc(&p.dyn, 8))) return rc;
  MMSSL_HIP_TRY(hipMemset(p.dyn, 0, 8 * sizeof(int32_t)));
  MMSSL_HIP_TRY(hipMemset(p.rowptr, 0, ((size_t)rows + 1) * sizeof(int32_t)));
  return 0;
}

}  // namespace

struct mmssl_graph_pair {
  mmssl_graph ui, iu;            // ui.fwd: users x items (by user), ui.bwd: its transpose; iu.fwd: items x users, ...
  unsigned long long* by_u = nullptr;     // sorted keys (user, item)
  unsigned long long* by_i = nullptr;     // sorted keys (item, user)
  int32_t n_users = 0, n_items = 0;
  int cap = 0;
};

extern "C" int mmssl_graph_pair_create(int32_t n_users, int32_t n_items, int64_t capacity, mmssl_graph_pair** out) {
  if (!out || n_users <= 0 || n_items <= 0 || capacity <= 0) return MMSSL_E_BADARG;
  *out = nullptr;
  if (capacity > kMaxPairs) return MMSSL_E_UNSUPP;
  mmssl_graph_pair* h = new (std::nothrow) mmssl_graph_pair();
  if (!h) return (int)hipErrorOutOfMemory;
  h->n_users = n_users;
  h->n_items = n_items;
  h->cap = (int)capacity;
  int rc = alloc_dir(h->ui.fwd, n_users, n_items, h->cap);
  if (!rc) rc = alloc_dir(h->ui.bwd, n_items, n_users, h->cap);
  if (!rc) rc = alloc_dir(h->iu.fwd, n_items, n_users, h->cap);
  if (!rc) rc = alloc_dir(h->iu.bwd, n_users, n_items, h->cap);
  if (!rc) rc = dev_alloc(&h->by_u, (size_t)h->cap);
  if (!rc) rc = dev_alloc(&h->by_i, (size_t)h->cap);
  if (!rc) rc = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(pairs_sort_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kMaxPairs * 8);
  if (rc) {
    mmssl_graph_pair_destroy(h);
    return rc;
  }
  *out = h;
  return 0;
}

extern "C" int mmssl_graph_pair_destroy(mmssl_graph_pair* h) {
  if (!h) return 0;
  free_dir(h->ui.fwd);
  free_dir(h->ui.bwd);
  free_dir(h->iu.fwd);
  free_dir(h->iu.bwd);
  if (h->by_u) (void)hipFree(h->by_u);
  if (h->by_i) (void)hipFree(h->by_i);
  delete h;
  return 0;
}

extern "C" int mmssl_graph_pair_get(mmssl_graph_pair* h, mmssl_graph** ui, mmssl_graph** iu) {
  if (!h || !ui || !iu) return MMSSL_E_BADARG;
  *ui = &h->ui;
  *iu = &h->iu;
  return 0;
}

extern "C" int mmssl_graph_pair_rebuild(mmssl_graph_pair* h, const int64_t* users, const int64_t* items, int64_t n,
                                        void* stream) {
  if (!h || n < 0 || n > h->cap || (n > 0 && (!users || !items))) return MMSSL_E_BADARG;
  hipStream_t s = as_stream(stream);
  const int ni = (int)n;
  int m = 1;
  while (m < ni) m <<= 1;
  DirPlan* dirs[4] = {&h->ui.fwd, &h->iu.bwd, &h->iu.fwd, &h->ui.bwd};
  hipLaunchKernelGGL(zero_counts_kernel, dim3(1), dim3(64), 0, s, dirs[0]->dyn, dirs[1]->dyn, dirs[2]->dyn, dirs[3]->dyn);
  MMSSL_LAUNCH_CHECK();
  if (ni > 0) {
    hipLaunchKernelGGL(pairs_sort_kernel, dim3(1), dim3(kSortThreads), (size_t)m * 8, s, users, items, ni, h->by_u);
    MMSSL_LAUNCH_CHECK();
    hipLaunchKernelGGL(pairs_sort_kernel, dim3(1), dim3(kSortThreads), (size_t)m * 8, s, items, users, ni, h->by_i);
    MMSSL_LAUNCH_CHECK();
  }
  const unsigned gu = (unsigned)((h->n_users + 1 + kBlock - 1) / kBlock), gi = (unsigned)((h->n_items + 1 + kBlock - 1) / kBlock);
  // row pointers of both groupings (the two directions that share a grouping share the row pointers' VALUES)
  hipLaunchKernelGGL(rowptr_kernel, dim3(gu), dim3(kBlock), 0, s, h->by_u, ni, h->n_users, h->ui.fwd.rowptr);
  hipLaunchKernelGGL(rowptr_kernel, dim3(gu), dim3(kBlock), 0, s, h->by_u, ni, h->n_users, h->iu.bwd.rowptr);
  hipLaunchKernelGGL(rowptr_kernel, dim3(gi), dim3(kBlock), 0, s, h->by_i, ni, h->n_items, h->iu.fwd.rowptr);
  hipLaunchKernelGGL(rowptr_kernel, dim3(gi), dim3(kBlock), 0, s, h->by_i, ni, h->n_items, h->ui.bwd.rowptr);
  MMSSL_LAUNCH_CHECK();
  if (ni > 0) {
    const unsigned ge = (unsigned)((ni + kBlock - 1) / kBlock);
    // by user: A_ui rows (weight by user degree) and A_iu^T rows (weight by item degree)
    hipLaunchKernelGGL(edges_kernel, dim3(ge), dim3(kBlock), 0, s, h->by_u, ni, h->ui.fwd.rowptr, h->iu.fwd.rowptr,
                       h->ui.fwd.edges, h->iu.bwd.edges);
    // by item: A_iu rows (weight by item degree) and A_ui^T rows (weight by user degree)
    hipLaunchKernelGGL(edges_kernel, dim3(ge), dim3(kBlock), 0, s, h->by_i, ni, h->iu.fwd.rowptr, h->ui.fwd.rowptr,
                       h->iu.fwd.edges, h->ui.bwd.edges);
    MMSSL_LAUNCH_CHECK();
  }
  for (int k = 0; k < 4; ++k) {
    DirPlan& p = *dirs[k];
    const unsigned gr = (unsigned)((p.rows + kBlock - 1) / kBlock);
    hipLaunchKernelGGL(plan_pass1_kernel, dim3(gr), dim3(kBlock), 0, s, p.rowptr, p.rows, p.gitems, p.witems, p.multi,
                       p.slot2multi, p.dyn);
    hipLaunchKernelGGL(plan_pass2_kernel, dim3(gr), dim3(kBlock), 0, s, p.rowptr, p.rows, p.witems, p.dyn);
    hipLaunchKernelGGL(plan_finish_kernel, dim3(1), dim3(1), 0, s, p.dyn, ni);
    MMSSL_LAUNCH_CHECK();
  }
  return 0;
}
