// Graph plan + CSR SpMM for gfx950 (MI355X).
//
// Replaces the reference's torch sparse COO handles and torch.sparse.mm call sites
// (/root/reference/MMSSL/main.py:89-112,513-520; Models.py:69-73,177-186,201-211).
//
// Design (HBM/L2-bound gather, NOT reshaped into a GEMM):
//   * d fp32 features per row are covered by LPR = d/4 lanes holding one float4 each, so a
//     neighbour-row read is ONE coalesced 16-B-per-lane request (d=64: 16 lanes = 256 B);
//     a wave64 therefore works on 64/LPR rows (or row slices) at once.
//   * the {col,val} pairs of a row tile are read once, coalesced (lane j takes edge j of
//     the tile), and broadcast inside the lane group with ds_bpermute; X gathers are
//     issued 4 deep per lane.
//   * load balance comes from the host-built plan: rows with <= kShortMax nnz are "group
//     items" sorted by degree (longest first: LPT order, and the 4 rows sharing a wave
//     have equal trip counts); longer rows are cut into wave items of <= kTaskNnz nnz whose
//     partial sums are combined in a fixed order (bitwise reproducible, no float atomics).
//   * the row softmax of the last GCN layer is fused into the store.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "graph_internal.hpp"

using namespace mmssl;

namespace {

// Work-list shaping of the graph plan (swept in round 2, tools/kbench.py at that revision): rows of at most 32 nonzeros
// are "short" (lane-group items), longer rows are cut into tasks of 128 nonzeros, items sorted by length.
constexpr int short_max() { return 32; }
constexpr int task_nnz() { return 128; }
constexpr int plan_sort() { return 1; }
// Rows that span several heavy blocks are combined by the last-arriving block inside the SpMM kernel (write-through
// partials + arrival ticket). The second-kernel form measured slower with the block-grouped plan (76.7 vs 75.2 us for the
// 6-SpMM forward chain, 16.3 vs 12.8 us for one eager launch) and was removed.

}  // namespace

namespace mmssl {
void free_dir(DirPlan& p) {
  if (p.dyn) (void)hipFree(p.dyn);
  if (p.bands) (void)hipFree(p.bands);
  if (p.wmap) (void)hipFree(p.wmap);
  if (p.rowptr) (void)hipFree(p.rowptr);
  if (p.edges) (void)hipFree(p.edges);
  if (p.gitems) (void)hipFree(p.gitems);
  if (p.witems) (void)hipFree(p.witems);
  if (p.multi) (void)hipFree(p.multi);
  if (p.slot2multi) (void)hipFree(p.slot2multi);
  p = DirPlan();
}
}  // namespace mmssl

namespace {

template <typename T>
int upload(T** dst, const T* src, size_t n) {
  *dst = nullptr;
  MMSSL_HIP_TRY(hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T)));
  if (n) MMSSL_HIP_TRY(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

// xcd_bands: 0 = band the group items when the graph has column locality (score >= kBandAuto), 1 = always, -1 = never
// Automatic banding wants clear locality AND balanced bands. The score of uniformly random columns is 1 / kBands only for
// long rows: a 3-edge row has a third of its edges in its "dominant" band by construction, so the Baby-shaped synthetic
// graph (mean degree 7) scores 0.24 - 0.33 without any locality; a community-structured graph scores > 0.9.
constexpr double kBandAuto = 0.50;
constexpr double kBandImbalance = 1.5;  // longest band / mean band: beyond this the idle blocks cost more than the locality buys

int build_dir(DirPlan& p, const int32_t* rowptr, const int32_t* col, const float* val, int32_t rows,
              int32_t cols, int64_t nnz, int xcd_bands, const int32_t* band_given = nullptr) {
  p.rows = rows;
  p.cols = cols;
  p.nnz = nnz;
  int64_t c[4];
  int rc = mmssl_plan_count_host(rowptr, rows, c);
  if (rc) return rc;
  p.n_g = c[0]; p.n_w = c[1]; p.n_multi = c[2]; p.n_slots = c[3];
  std::vector<int32_t> g((size_t)p.n_g * 4), w((size_t)p.n_w * 4), m((size_t)p.n_multi * 4);
  rc = mmssl_plan_fill_host(rowptr, rows, g.data(), w.data(), m.data());
  if (rc) return rc;
  if ((xcd_bands >= 0 || band_given) && p.n_g > 0 && nnz > 0) {
    std::vector<int32_t> band((size_t)rows);
    double score = 0.0;
    if (band_given) {              // the caller clustered the graph (GraphPlan: co-clustering of rows and columns)
      for (int32_t r = 0; r < rows; ++r) {
        if (band_given[r] < 0 || band_given[r] >= kBands) return MMSSL_E_BADARG;
        band[(size_t)r] = band_given[r];
      }
      xcd_bands = 1;
    } else {
      rc = mmssl_plan_band_host(rowptr, col, rows, cols, kBands, band.data(), &score);
      if (rc) return rc;
    }
    p.band_score = score;
    if (xcd_bands > 0 || score >= kBandAuto) {
      std::vector<int32_t> banded(g);
      int32_t start[kBands + 1];
      rc = mmssl_plan_band_group_items_host(banded.data(), p.n_g, band.data(), kBands, start);
      if (rc) return rc;
      int64_t longest = 0;
      for (int x = 0; x < kBands; ++x) longest = std::max<int64_t>(longest, start[x + 1] - start[x]);
      if (xcd_bands > 0 || (double)longest * kBands <= kBandImbalance * (double)p.n_g) {
        g.swap(banded);
        p.band_max = longest;
        if ((rc = upload(&p.bands, start, (size_t)kBands + 1))) return rc;
        if (p.n_w > 0) {
          std::vector<int32_t> wmap((size_t)((p.n_w + 3) / 4));
          rc = mmssl_plan_band_wave_blocks_host(w.data(), p.n_w, band.data(), kBands, wmap.data());
          if (rc) return rc;
          if ((rc = upload(&p.wmap, wmap.data(), wmap.size()))) return rc;
        }
      }
    }
  }
  std::vector<Edge> e((size_t)nnz);
  for (int64_t i = 0; i < nnz; ++i) { e[i].col = col[i]; e[i].val = val[i]; }
  if ((rc = upload(&p.rowptr, rowptr, (size_t)rows + 1))) return rc;
  if ((rc = upload(&p.edges, e.data(), (size_t)nnz))) return rc;
  if ((rc = upload(&p.gitems, reinterpret_cast<const int4*>(g.data()), (size_t)p.n_g))) return rc;
  if ((rc = upload(&p.witems, reinterpret_cast<const int4*>(w.data()), (size_t)p.n_w))) return rc;
  if ((rc = upload(&p.multi, reinterpret_cast<const int4*>(m.data()), (size_t)p.n_multi))) return rc;
  std::vector<int32_t> s2m((size_t)p.n_slots);
  for (int64_t k = 0; k < p.n_multi; ++k)
    for (int32_t j = 0; j < m[k * 4 + 2]; ++j) s2m[(size_t)m[k * 4 + 1] + j] = (int32_t)k;
  if ((rc = upload(&p.slot2multi, s2m.data(), (size_t)p.n_slots))) return rc;
  return 0;
}

}  // namespace


// ======================================================================================
// host planning (pure CPU; exported for CPU-only tests)
// ======================================================================================
extern "C" int mmssl_csr_validate_host(const int32_t* rowptr, const int32_t* col, int32_t rows,
                                       int32_t cols, int64_t nnz) {
  if (rows < 0 || cols < 0 || nnz < 0 || !rowptr) return MMSSL_E_BADARG;
  if (nnz > 0 && !col) return MMSSL_E_BADARG;
  if (nnz > INT32_MAX) return MMSSL_E_UNSUPP;
  if (rowptr[0] != 0 || rowptr[rows] != nnz) return MMSSL_E_BADARG;
  for (int32_t r = 0; r < rows; ++r)
    if (rowptr[r + 1] < rowptr[r]) return MMSSL_E_BADARG;
  for (int64_t i = 0; i < nnz; ++i)
    if (col[i] < 0 || col[i] >= cols) return MMSSL_E_BADARG;
  return 0;
}

// Stable counting sort by column: within a transposed row, entries keep ascending source-row
// order, so the backward's summation order is a pure function of the matrix.
extern "C" int mmssl_csr_transpose_host(const int32_t* rowptr, const int32_t* col, const float* val,
                                        int32_t rows, int32_t cols, int64_t nnz, int32_t* t_rowptr,
                                        int32_t* t_col, float* t_val) {
  if (!rowptr || !t_rowptr || (nnz > 0 && (!col || !val || !t_col || !t_val))) return MMSSL_E_BADARG;
  std::fill(t_rowptr, t_rowptr + cols + 1, 0);
  for (int64_t i = 0; i < nnz; ++i) t_rowptr[col[i] + 1]++;
  for (int32_t c = 0; c < cols; ++c) t_rowptr[c + 1] += t_rowptr[c];
  std::vector<int32_t> cur(t_rowptr, t_rowptr + cols);
  for (int32_t r = 0; r < rows; ++r)
    for (int32_t i = rowptr[r]; i < rowptr[r + 1]; ++i) {
      const int32_t dst = cur[col[i]]++;
      t_col[dst] = r;
      t_val[dst] = val[i];
    }
  return 0;
}

// Work-list layout (see include/mmssl_hip.h):
//   group items : rows with deg <= short_max, one lane group each, longest first.
//   wave items  : <= task_nnz-edge slices {row, beg, end, code}; the list is [heavy section | light section].
//     light  (code -1): a whole row of short_max < deg <= task_nnz; four unrelated rows share a block.
//     heavy  : rows with deg > task_nnz, heaviest first; a row's slices are padded with no-work items
//              {-1, 0, 0, code} to a multiple of 4, so every BLOCK of the heavy section holds slices of ONE row.
//              Its four waves add their partial sums through LDS;
//              code -2 : the row fits this block (deg <= 4 task_nnz) -> epilogue + store, no partial slot;
//              code s>=0: the row spans several blocks; this block owns partial slot s (one per block, not
//                         one per slice: 4x fewer slots and 4x fewer rows that need a cross-block combine).
//   multi       : rows spanning several blocks {row, first_slot, n_slots, 0}.
namespace {
struct HeavyRow { int32_t row; int32_t deg; };
inline void heavy_rows_sorted(const int32_t* rowptr, int32_t rows, int tn, std::vector<HeavyRow>& out) {
  for (int32_t r = 0; r < rows; ++r) {
    const int32_t deg = rowptr[r + 1] - rowptr[r];
    if (deg > tn) out.push_back({r, deg});
  }
  if (plan_sort())
    std::stable_sort(out.begin(), out.end(), [](const HeavyRow& a, const HeavyRow& b) { return a.deg > b.deg; });
}
}  // namespace

extern "C" int mmssl_plan_count_host(const int32_t* rowptr, int32_t rows, int64_t counts[4]) {
  if (!rowptr || !counts || rows < 0) return MMSSL_E_BADARG;
  const int smax = short_max(), tn = task_nnz();
  int64_t g = 0, w = 0, m = 0, s = 0;
  for (int32_t r = 0; r < rows; ++r) {
    const int64_t deg = rowptr[r + 1] - rowptr[r];
    if (deg <= smax) {
      ++g;
    } else if (deg <= tn) {
      ++w;                                   // light wave item
    } else {
      const int64_t t = (deg + tn - 1) / tn;   // slices
      const int64_t nb = (t + 3) / 4;          // blocks
      w += nb * 4;                             // padded
      if (nb > 1) { ++m; s += nb; }
    }
  }
  counts[0] = g; counts[1] = w; counts[2] = m; counts[3] = s;
  return 0;
}

extern "C" int mmssl_plan_fill_host(const int32_t* rowptr, int32_t rows, int32_t* gi, int32_t* wi,
                                    int32_t* mi) {
  if (!rowptr || rows < 0) return MMSSL_E_BADARG;
  const int smax = short_max(), tn = task_nnz();
  // group items: counting sort by degree, longest first (stable: ascending row id per degree)
  std::vector<int64_t> start(smax + 2, 0);
  if (plan_sort()) {
    for (int32_t r = 0; r < rows; ++r) {
      const int deg = rowptr[r + 1] - rowptr[r];
      if (deg <= smax) start[smax - deg + 1]++;
    }
    for (int k = 0; k <= smax; ++k) start[k + 1] += start[k];
  }
  int64_t gseq = 0, w = 0, m = 0, slot = 0;
  // heavy section first
  std::vector<HeavyRow> heavy;
  heavy_rows_sorted(rowptr, rows, tn, heavy);
  for (const HeavyRow& hr : heavy) {
    const int32_t beg = rowptr[hr.row], end = rowptr[hr.row + 1];
    const int32_t t = (hr.deg + tn - 1) / tn, nb = (t + 3) / 4;
    if (nb > 1) {
      int32_t* mm = mi + m * 4;
      mm[0] = hr.row; mm[1] = (int32_t)slot; mm[2] = nb; mm[3] = 0;
      ++m;
    }
    for (int32_t k = 0; k < nb * 4; ++k) {
      int32_t* it = wi + w * 4;
      const int32_t code = nb > 1 ? (int32_t)(slot + k / 4) : -2;
      if (k < t) {
        it[0] = hr.row;
        it[1] = beg + k * tn;
        it[2] = std::min(end, beg + (k + 1) * tn);
      } else {
        it[0] = -1; it[1] = 0; it[2] = 0;
      }
      it[3] = code;
      ++w;
    }
    if (nb > 1) slot += nb;
  }
  for (int32_t r = 0; r < rows; ++r) {
    const int32_t beg = rowptr[r], end = rowptr[r + 1];
    const int deg = end - beg;
    if (deg <= smax) {
      const int64_t pos = plan_sort() ? start[smax - deg]++ : gseq++;
      int32_t* it = gi + pos * 4;
      it[0] = r; it[1] = beg; it[2] = end; it[3] = -1;
    } else if (deg <= tn) {
      int32_t* it = wi + w * 4;
      it[0] = r; it[1] = beg; it[2] = end; it[3] = -1;
      ++w;
    }
  }
  return 0;
}

// XCD banding (see graph_internal.hpp): band_of_row[r] = the column band (cols cut into n_bands equal ranges) most of row
// r's edges fall into (ties: the lowest band; empty rows: r % n_bands); *score = the fraction of all edges that fall into
// their row's band - 1 / n_bands for uniformly random columns, towards 1 for a graph whose rows reference a narrow range.
extern "C" int mmssl_plan_band_host(const int32_t* rowptr, const int32_t* col, int32_t rows, int32_t cols,
                                    int32_t n_bands, int32_t* band_of_row, double* score) {
  if (!rowptr || !band_of_row || rows < 0 || cols < 0 || n_bands < 1 || n_bands > 64) return MMSSL_E_BADARG;
  if (rowptr[rows] > 0 && !col) return MMSSL_E_BADARG;
  const int64_t width = std::max<int64_t>(1, ((int64_t)cols + n_bands - 1) / n_bands);
  int64_t hit = 0, total = 0;
  std::vector<int32_t> cnt((size_t)n_bands);
  for (int32_t r = 0; r < rows; ++r) {
    std::fill(cnt.begin(), cnt.end(), 0);
    for (int32_t i = rowptr[r]; i < rowptr[r + 1]; ++i) cnt[(size_t)std::min<int64_t>(col[i] / width, n_bands - 1)]++;
    int best = 0;
    for (int b = 1; b < n_bands; ++b)
      if (cnt[b] > cnt[best]) best = b;
    const int32_t deg = rowptr[r + 1] - rowptr[r];
    band_of_row[r] = deg > 0 ? best : r % n_bands;
    hit += cnt[best];
    total += deg;
  }
  if (score) *score = total > 0 ? (double)hit / (double)total : 0.0;
  return 0;
}

// Stable partition of the (degree-sorted) group items by their row's band: band-major, longest first inside a band.
// band_start[x] .. band_start[x + 1] = the items of band x.
extern "C" int mmssl_plan_band_group_items_host(int32_t* group_items, int64_t n_g, const int32_t* band_of_row,
                                                int32_t n_bands, int32_t* band_start) {
  if (n_g < 0 || (n_g > 0 && (!group_items || !band_of_row)) || !band_start || n_bands < 1 || n_bands > 64)
    return MMSSL_E_BADARG;
  std::vector<int64_t> cnt((size_t)n_bands + 1, 0);
  for (int64_t k = 0; k < n_g; ++k) {
    const int32_t b = band_of_row[group_items[k * 4]];
    if (b < 0 || b >= n_bands) return MMSSL_E_BADARG;
    cnt[(size_t)b + 1]++;
  }
  for (int b = 0; b < n_bands; ++b) cnt[(size_t)b + 1] += cnt[(size_t)b];
  for (int b = 0; b <= n_bands; ++b) band_start[b] = (int32_t)cnt[(size_t)b];
  std::vector<int32_t> out((size_t)n_g * 4);
  std::vector<int64_t> cur(cnt.begin(), cnt.end() - 1);
  for (int64_t k = 0; k < n_g; ++k) {
    const int64_t dst = cur[(size_t)band_of_row[group_items[k * 4]]]++;
    std::memcpy(&out[(size_t)dst * 4], &group_items[k * 4], 4 * sizeof(int32_t));
  }
  std::memcpy(group_items, out.data(), out.size() * sizeof(int32_t));
  return 0;
}

// Wave blocks (consecutive groups of 4 wave items): the light section (whole rows, code -1, behind the heavy section) is
// reordered band-major in place, then every hardware block b gets a wave block of band b % n_bands while that band has
// any (in list order: heaviest first), else one of the band with the most blocks left. wmap[b] = the wave block b runs.
extern "C" int mmssl_plan_band_wave_blocks_host(int32_t* wave_items, int64_t n_w, const int32_t* band_of_row,
                                                int32_t n_bands, int32_t* wmap) {
  if (n_w < 0 || (n_w > 0 && (!wave_items || !band_of_row || !wmap)) || n_bands < 1 || n_bands > 64) return MMSSL_E_BADARG;
  const int64_t nb = (n_w + 3) / 4;
  int64_t light0 = n_w;                                   // first light item: a real row with code -1
  for (int64_t k = 0; k < n_w; ++k)
    if (wave_items[k * 4 + 3] == -1 && wave_items[k * 4] >= 0) { light0 = k; break; }
  if (light0 % 4) return MMSSL_E_BADARG;                  // the heavy section is padded to whole blocks
  {
    std::vector<int32_t> tmp((size_t)(n_w - light0) * 4);
    int64_t dst = 0;
    for (int b = 0; b < n_bands; ++b)
      for (int64_t k = light0; k < n_w; ++k)
        if (band_of_row[wave_items[k * 4]] == b) {
          std::memcpy(&tmp[(size_t)dst * 4], &wave_items[k * 4], 4 * sizeof(int32_t));
          ++dst;
        }
    if (dst != n_w - light0) return MMSSL_E_BADARG;
    if (!tmp.empty()) std::memcpy(&wave_items[light0 * 4], tmp.data(), tmp.size() * sizeof(int32_t));
  }
  std::vector<std::vector<int32_t>> q((size_t)n_bands);
  for (int64_t vb = 0; vb < nb; ++vb) {
    const int32_t row = wave_items[vb * 16];              // first item of the block: never padding
    if (row < 0) return MMSSL_E_BADARG;
    q[(size_t)band_of_row[row]].push_back((int32_t)vb);
  }
  std::vector<size_t> head((size_t)n_bands, 0);
  for (int64_t b = 0; b < nb; ++b) {
    int x = (int)(b % n_bands);
    if (head[(size_t)x] >= q[(size_t)x].size()) {         // this band is done: help the fullest one
      size_t best = 0;
      for (int y = 0; y < n_bands; ++y) {
        const size_t left = q[(size_t)y].size() - head[(size_t)y];
        if (left > best) { best = left; x = y; }
      }
    }
    wmap[b] = q[(size_t)x][head[(size_t)x]++];
  }
  return 0;
}

// ======================================================================================
// graph object
// ======================================================================================
extern "C" int mmssl_graph_create(const int32_t* rowptr, const int32_t* col, const float* val,
                                  int32_t rows, int32_t cols, int64_t nnz, void* stream,
                                  mmssl_graph** out) {
  return mmssl_graph_create_ex(rowptr, col, val, rows, cols, nnz, 0, stream, out);
}

extern "C" int mmssl_graph_create_ex(const int32_t* rowptr, const int32_t* col, const float* val,
                                     int32_t rows, int32_t cols, int64_t nnz, int xcd_bands, void* stream,
                                     mmssl_graph** out) {
  return mmssl_graph_create_banded(rowptr, col, val, rows, cols, nnz, xcd_bands, nullptr, nullptr, stream, out);
}

extern "C" int mmssl_graph_create_banded(const int32_t* rowptr, const int32_t* col, const float* val,
                                         int32_t rows, int32_t cols, int64_t nnz, int xcd_bands, const int32_t* row_band,
                                         const int32_t* col_band, void* stream, mmssl_graph** out) {
  (void)stream;
  if ((row_band == nullptr) != (col_band == nullptr)) return MMSSL_E_BADARG;  // set-up is synchronous (hipMemcpy); the handle is usable on any stream afterwards
  if (!out) return MMSSL_E_BADARG;
  *out = nullptr;
  int rc = mmssl_csr_validate_host(rowptr, col, rows, cols, nnz);
  if (rc) return rc;
  if (nnz > 0 && !val) return MMSSL_E_BADARG;
  std::vector<int32_t> t_rowptr((size_t)cols + 1), t_col((size_t)nnz);
  std::vector<float> t_val((size_t)nnz);
  rc = mmssl_csr_transpose_host(rowptr, col, val, rows, cols, nnz, t_rowptr.data(), t_col.data(),
                                t_val.data());
  if (rc) return rc;
  mmssl_graph* g = new (std::nothrow) mmssl_graph();
  if (!g) return (int)hipErrorOutOfMemory;
  rc = build_dir(g->fwd, rowptr, col, val, rows, cols, nnz, xcd_bands, row_band);
  if (!rc) rc = build_dir(g->bwd, t_rowptr.data(), t_col.data(), t_val.data(), cols, rows, nnz, xcd_bands, col_band);
  if (rc) {
    free_dir(g->fwd);
    free_dir(g->bwd);
    delete g;
    return rc;
  }
  *out = g;
  return 0;
}

extern "C" int mmssl_graph_destroy(mmssl_graph* g) {
  if (!g) return 0;
  free_dir(g->fwd);
  free_dir(g->bwd);
  delete g;
  return 0;
}

extern "C" int mmssl_graph_info(const mmssl_graph* g, int64_t info[16]) {
  if (!g || !info) return MMSSL_E_BADARG;
  std::memset(info, 0, 16 * sizeof(int64_t));
  info[0] = g->fwd.rows; info[1] = g->fwd.cols; info[2] = g->fwd.nnz;
  info[3] = g->fwd.n_g; info[4] = g->fwd.n_w; info[5] = g->fwd.n_multi; info[6] = g->fwd.n_slots;
  info[8] = g->bwd.n_g; info[9] = g->bwd.n_w; info[10] = g->bwd.n_multi; info[11] = g->bwd.n_slots;
  info[12] = short_max(); info[13] = task_nnz(); info[14] = plan_sort();
  // XCD banding: bit 0 / 1 = the forward / transposed direction's group items are banded; scores in 1/1000
  info[15] = (g->fwd.bands ? 1 : 0) | (g->bwd.bands ? 2 : 0) | ((int64_t)(g->fwd.band_score * 1000.0 + 0.5) << 8) |
             ((int64_t)(g->bwd.band_score * 1000.0 + 0.5) << 24);
  return 0;
}

// (rowptr, col, val) of either direction to the host; `cap` = capacity of col / val (entries). Returns the number
// of stored entries through *nnz_out (device-built plans keep it in device memory). Synchronises the stream.
extern "C" int mmssl_graph_export_f32(const mmssl_graph* g, int transpose, int32_t* rowptr, int32_t* col, float* val,
                                      int64_t cap, int64_t* nnz_out, void* stream) {
  if (!g || !rowptr || !nnz_out) return MMSSL_E_BADARG;
  MMSSL_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
  const DirPlan& p = transpose ? g->bwd : g->fwd;
  MMSSL_HIP_TRY(hipMemcpy(rowptr, p.rowptr, ((size_t)p.rows + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
  const int64_t nnz = rowptr[p.rows];
  *nnz_out = nnz;
  if (nnz > cap) return MMSSL_E_WORKSPACE;
  if (nnz) {
    if (!col || !val) return MMSSL_E_BADARG;
    std::vector<Edge> e((size_t)nnz);
    MMSSL_HIP_TRY(hipMemcpy(e.data(), p.edges, (size_t)nnz * sizeof(Edge), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < nnz; ++i) { col[i] = e[i].col; val[i] = e[i].val; }
  }
  return 0;
}

extern "C" int mmssl_graph_export_transpose(const mmssl_graph* g, int32_t* t_rowptr, int32_t* t_col,
                                            float* t_val, void* stream) {
  if (!g || !t_rowptr) return MMSSL_E_BADARG;
  MMSSL_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
  const DirPlan& p = g->bwd;
  MMSSL_HIP_TRY(hipMemcpy(t_rowptr, p.rowptr, ((size_t)p.rows + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (p.nnz) {
    if (!t_col || !t_val) return MMSSL_E_BADARG;
    std::vector<Edge> e((size_t)p.nnz);
    MMSSL_HIP_TRY(hipMemcpy(e.data(), p.edges, (size_t)p.nnz * sizeof(Edge), hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < p.nnz; ++i) { t_col[i] = e[i].col; t_val[i] = e[i].val; }
  }
  return 0;
}

// ======================================================================================
// kernels
// ======================================================================================
namespace {

// Accumulate sum_e val[e] * X[col[e], :] over edges [beg,end), visiting LPR-edge tiles
// first_tile, first_tile+tile_stride, ...  One lane group (LPR lanes) per call; `lig` = lane in
// group. Each lane holds 4 consecutive features (one float4).
// LD: X rows are `ldx4` float4s apart (a column chunk of a wider row-major table) instead of LPR.
template <int LPR, bool LD = false>
__device__ __forceinline__ float4 gather_rows(const Edge* __restrict__ edges,
                                              const float4* __restrict__ X, int beg, int end,
                                              int first_tile, int tile_stride, int lig, int ldx4 = LPR) {
  const size_t px = LD ? (size_t)ldx4 : (size_t)LPR;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const int n = end - beg;
  for (int b0 = first_tile * LPR; b0 < n; b0 += tile_stride * LPR) {
    const int e = b0 + lig;
    Edge my;
    my.col = -1;
    my.val = 0.f;
    if (e < n) my = edges[beg + e];            // coalesced: LPR consecutive 8-B pairs
    // lanes past the end multiply the tile's FIRST column (one this row references anyway)
    // by 0, so the 4-deep unrolled loop needs no per-load predicate.
    const int pad = __shfl(my.col, 0, LPR);
    if (my.col < 0) my.col = pad;
    const int cnt = min(LPR, n - b0);
    for (int k = 0; k < cnt; k += 4) {
      const int c0 = __shfl(my.col, k + 0, LPR), c1 = __shfl(my.col, k + 1, LPR);
      const int c2 = __shfl(my.col, k + 2, LPR), c3 = __shfl(my.col, k + 3, LPR);
      const float v0 = __shfl(my.val, k + 0, LPR), v1 = __shfl(my.val, k + 1, LPR);
      const float v2 = __shfl(my.val, k + 2, LPR), v3 = __shfl(my.val, k + 3, LPR);
      const float4 x0 = X[(size_t)c0 * px + lig];
      const float4 x1 = X[(size_t)c1 * px + lig];
      const float4 x2 = X[(size_t)c2 * px + lig];
      const float4 x3 = X[(size_t)c3 * px + lig];
      acc = f4_fma(v0, x0, acc);
      acc = f4_fma(v1, x1, acc);
      acc = f4_fma(v2, x2, acc);
      acc = f4_fma(v3, x3, acc);
    }
  }
  return acc;
}

// Store-side epilogues (row-local, fused into the SpMM):
//   NONE              y = acc
//   SOFTMAX           y = softmax(acc)                                   (last GCN layer, fwd)
//   AXPY              y = acc + alpha * Z[row]                           (bwd: + layer-mean grad)
//   AXPY_SOFTMAX_BWD  t = acc + alpha * Z[row];  y = S[row] * (t - <t, S[row]>)   (bwd through softmax)
//   MASK              y = keep[m][row][c] ? acc * alpha : 0   (dropout backward fused into the SpMM that produces the
//                     projection's output gradient; Y packs nm modalities of dm features side by side: column
//                     m * dm + c of row `row` reads keep[(m * rows + row) * dm + c], the [nm, rows, dm] mask layout)
struct EpiArgs {
  const float4* Z;
  const float4* S;
  float alpha;
  const uint8_t* keep;
  int dm;
  int64_t rows;
  int ldx4, ldy4;      // LD kernels: row pitch (in float4) of X and of Y / Z / S
};

template <int LPR, int EPI, bool LD = false>
__device__ __forceinline__ float4 apply_epilogue(float4 acc, int row, int lig, const EpiArgs& e) {
  const size_t py = LD ? (size_t)e.ldy4 : (size_t)LPR;
  if (EPI == MMSSL_EPI_SOFTMAX) return row_softmax<LPR>(acc);
  if (EPI == MMSSL_EPI_MASK) {
    const int col = 4 * lig, m = col / e.dm;
    const uint32_t k4 = *reinterpret_cast<const uint32_t*>(e.keep + ((int64_t)m * e.rows + row) * e.dm + (col - m * e.dm));
    acc.x = (k4 & 0x000000ffu) ? acc.x * e.alpha : 0.f;
    acc.y = (k4 & 0x0000ff00u) ? acc.y * e.alpha : 0.f;
    acc.z = (k4 & 0x00ff0000u) ? acc.z * e.alpha : 0.f;
    acc.w = (k4 & 0xff000000u) ? acc.w * e.alpha : 0.f;
    return acc;
  }
  if (EPI == MMSSL_EPI_AXPY || EPI == MMSSL_EPI_AXPY_SOFTMAX_BWD) {
    const float4 z = e.Z[(size_t)row * py + lig];
    acc.x = fmaf(e.alpha, z.x, acc.x);
    acc.y = fmaf(e.alpha, z.y, acc.y);
    acc.z = fmaf(e.alpha, z.z, acc.z);
    acc.w = fmaf(e.alpha, z.w, acc.w);
  }
  if (EPI == MMSSL_EPI_AXPY_SOFTMAX_BWD) {
    const float4 y = e.S[(size_t)row * py + lig];
    const float dot = group_sum<LPR>(f4_dot(acc, y));
    acc = make_float4(y.x * (acc.x - dot), y.y * (acc.y - dot), y.z * (acc.z - dot), y.w * (acc.w - dot));
  }
  return acc;
}

// grid = n_wblocks (4 wave items each, heaviest first) + n_gblocks (256/LPR group items each)
template <int LPR, int EPI, bool LD = false>
__global__ __launch_bounds__(kBlock) void spmm_kernel(const int4* __restrict__ gitems, int n_g,
                                                      const int4* __restrict__ witems, int n_w,
                                                      int n_wblocks, const Edge* __restrict__ edges,
                                                      const float4* __restrict__ X,
                                                      float4* __restrict__ Y,
                                                      float4* __restrict__ partials, EpiArgs epi,
                                                      const int4* __restrict__ multi,
                                                      const int32_t* __restrict__ slot2multi,
                                                      int32_t* __restrict__ arrivals,
                                                      const int32_t* __restrict__ dyn,
                                                      const int32_t* __restrict__ bands,
                                                      const int32_t* __restrict__ wmap) {
  // device-built plans (csrc/graphdev.hip): the item counts live in device memory and the grid is an upper bound
  if (dyn) {
    n_g = dyn[0];
    n_w = dyn[1];
    n_wblocks = (n_w + 3) >> 2;
  }
  constexpr int GPW = kWave / LPR;   // lane groups per wave
  constexpr int GPB = kBlock / LPR;  // lane groups per block
  const int lane = threadIdx.x & 63;
  const int lig = lane & (LPR - 1);
  const size_t py = LD ? (size_t)epi.ldy4 : (size_t)LPR;
  if ((int)blockIdx.x >= n_wblocks) {
    int gi = ((int)blockIdx.x - n_wblocks) * GPB + (int)threadIdx.x / LPR;
    if (bands) {
      // XCD-banded plan: this block runs (observed placement, a speed matter only) on XCD blockIdx % 8 and takes the
      // j-th chunk of THAT band's items, so an XCD's L2 mostly sees one band of the gathered table
      const int x = (int)blockIdx.x & (kBands - 1);
      const int k = (int)blockIdx.x - n_wblocks;
      const int j = (k - ((x - n_wblocks) & (kBands - 1))) >> 3;
      gi = bands[x] + j * GPB + (int)threadIdx.x / LPR;
      if (gi >= bands[x + 1]) return;
    } else if (gi >= n_g) {
      return;
    }
    const int4 it = gitems[gi];
    float4 acc = gather_rows<LPR, LD>(edges, X, it.y, it.z, 0, 1, lig, epi.ldx4);
    acc = apply_epilogue<LPR, EPI, LD>(acc, it.x, lig, epi);
    Y[(size_t)it.x * py + lig] = acc;
  } else {
    const int wave = (int)threadIdx.x >> 6;
    const int vb = wmap ? wmap[blockIdx.x] : (int)blockIdx.x;    // XCD-banded plan: the wave block chosen for this block
    const int wi = vb * 4 + wave;
    const int code0 = witems[vb * 4].w;                    // block-uniform: light (-1) or heavy block
    if (code0 == -1) {                                      // four unrelated whole rows
      if (wi >= n_w) return;
      const int4 it = witems[wi];
      float4 acc = gather_rows<LPR, LD>(edges, X, it.y, it.z, lane / LPR, GPW, lig, epi.ldx4);
      acc.x = cross_group_sum<LPR>(acc.x);
      acc.y = cross_group_sum<LPR>(acc.y);
      acc.z = cross_group_sum<LPR>(acc.z);
      acc.w = cross_group_sum<LPR>(acc.w);
      acc = apply_epilogue<LPR, EPI, LD>(acc, it.x, lig, epi);
      if (lane < LPR) Y[(size_t)it.x * py + lig] = acc;
      return;
    }
    // ---- heavy block: up to four slices of ONE row (the heavy section is padded to whole blocks) ----
    __shared__ float4 red[4][LPR];
    const int4 it = witems[wi];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (it.x >= 0) {
      acc = gather_rows<LPR, LD>(edges, X, it.y, it.z, lane / LPR, GPW, lig, epi.ldx4);
      acc.x = cross_group_sum<LPR>(acc.x);
      acc.y = cross_group_sum<LPR>(acc.y);
      acc.z = cross_group_sum<LPR>(acc.z);
      acc.w = cross_group_sum<LPR>(acc.w);
    }
    if (lane < LPR) red[wave][lig] = acc;
    __syncthreads();
    if (wave != 0) return;
    const int row = witems[vb * 4].x;                       // the first slice of a block is never padding
    if (lane < LPR) {
      const float4 b1 = red[1][lig], b2 = red[2][lig], b3 = red[3][lig];
      acc.x = ((acc.x + b1.x) + b2.x) + b3.x;
      acc.y = ((acc.y + b1.y) + b2.y) + b3.y;
      acc.z = ((acc.z + b1.z) + b2.z) + b3.z;
      acc.w = ((acc.w + b1.w) + b2.w) + b3.w;
    }
    if (code0 == -2) {                                      // the whole row lives in this block
      acc = apply_epilogue<LPR, EPI, LD>(acc, row, lig, epi);
      if (lane < LPR) Y[(size_t)row * py + lig] = acc;
      return;
    }
    {
      const int slot = code0;
      // ---- in-kernel combine by the LAST-arriving block of this row (split-K arrival pattern) ----
      // The protocol below (write-through sc1 stores drained with vmcnt, relaxed ticket, sc1 re-reads; no
      // release/acquire fence) relies on gfx9 store accounting (stores retire through vmcnt) and on agent-scope
      // relaxed atomics lowering to sc1 accesses. Other targets must use a fence pair or the two-stage mode.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "spmm_kernel's in-kernel combine is written for gfx942/gfx950 memory semantics"
#endif
      // The 16*LPR-byte partial is stored WRITE-THROUGH (agent-scope relaxed atomic stores lower to
      // `global_store ... sc1`), drained, then one lane takes a ticket: no release fence, so the
      // other rows' dirty output lines stay in this XCD's L2. The last arriver re-reads every
      // slot with sc1 loads (bypass the non-coherent L1) in a FIXED order, so the result does not
      // depend on which block happens to be last.
      typedef unsigned long long u64;
      u64* pw = reinterpret_cast<u64*>(partials);
      if (lane < LPR) {
        u64 lo, hi;
        __builtin_memcpy(&lo, &acc.x, 8);
        __builtin_memcpy(&hi, &acc.z, 8);
        const size_t o = ((size_t)slot * LPR + lig) * 2;
        __hip_atomic_store(pw + o, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pw + o + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int m = slot2multi[slot];
      const int4 mr = multi[m];            // {row, first_slot, n_slots, 0}
      int last = 0;
      if (lane == 0) {
        const int prev = __hip_atomic_fetch_add(arrivals + m, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = (prev == mr.z - 1);
        if (last) __hip_atomic_store(arrivals + m, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // re-arm
      }
      last = __builtin_amdgcn_readfirstlane(last);
      if (!last) return;
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = lane / LPR; k < mr.z; k += GPW) {
        const size_t o = ((size_t)(mr.y + k) * LPR + lig) * 2;
        const u64 lo = __hip_atomic_load(pw + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 hi = __hip_atomic_load(pw + o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float2 a, b;
        __builtin_memcpy(&a, &lo, 8);
        __builtin_memcpy(&b, &hi, 8);
        sum.x += a.x;
        sum.y += a.y;
        sum.z += b.x;
        sum.w += b.y;
      }
      sum.x = cross_group_sum<LPR>(sum.x);
      sum.y = cross_group_sum<LPR>(sum.y);
      sum.z = cross_group_sum<LPR>(sum.z);
      sum.w = cross_group_sum<LPR>(sum.w);
      sum = apply_epilogue<LPR, EPI, LD>(sum, mr.x, lig, epi);
      if (lane < LPR) Y[(size_t)mr.x * py + lig] = sum;
    }
  }
}

// Workspace of one SpMM launch: [partial slots: n_slots * d floats | arrival counters: n_multi int32].
// The counters belong to the WORKSPACE, not to the plan: launches that overlap on different streams use
// different workspaces (GraphPlan.twin), and each must count its own arrivals. They must be zero before the
// first launch; every launch leaves them zero again (the last arriver re-arms its counter).
inline size_t ws_partials_bytes(const DirPlan& p, int d) { return (((size_t)p.n_slots * (size_t)d * sizeof(float)) + 15) & ~(size_t)15; }
inline size_t ws_total_bytes(const DirPlan& p, int d) {
  return ws_partials_bytes(p, d) + ((((size_t)p.n_multi * sizeof(int32_t)) + 15) & ~(size_t)15);
}

template <int LPR, int EPI, bool LD = false>
int launch_spmm(const DirPlan& p, const float* X, float* Y, float* partials, const EpiArgs& epi,
                hipStream_t s) {
  int32_t* arrivals = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(partials) + ws_partials_bytes(p, LPR * 4));
  constexpr int GPB = kBlock / LPR;
  const int n_wblocks = (int)((p.n_w + 3) / 4);
  const int n_gblocks = p.bands ? kBands * (int)((p.band_max + GPB - 1) / GPB) : (int)((p.n_g + GPB - 1) / GPB);
  if (n_wblocks + n_gblocks > 0) {
    hipLaunchKernelGGL((spmm_kernel<LPR, EPI, LD>), dim3(n_wblocks + n_gblocks), dim3(kBlock), 0, s,
                       p.gitems, (int)p.n_g, p.witems, (int)p.n_w, n_wblocks, p.edges,
                       reinterpret_cast<const float4*>(X), reinterpret_cast<float4*>(Y),
                       reinterpret_cast<float4*>(partials), epi, p.multi, p.slot2multi,
                       arrivals, p.dyn, p.bands, p.wmap);
    MMSSL_LAUNCH_CHECK();
  }
  return 0;
}

template <int LPR>
int dispatch_epi(const DirPlan& p, const float* X, float* Y, float* partials, int epi, const EpiArgs& e,
                 hipStream_t s) {
  switch (epi) {
    case MMSSL_EPI_NONE: return launch_spmm<LPR, MMSSL_EPI_NONE>(p, X, Y, partials, e, s);
    case MMSSL_EPI_SOFTMAX: return launch_spmm<LPR, MMSSL_EPI_SOFTMAX>(p, X, Y, partials, e, s);
    case MMSSL_EPI_AXPY: return launch_spmm<LPR, MMSSL_EPI_AXPY>(p, X, Y, partials, e, s);
    case MMSSL_EPI_AXPY_SOFTMAX_BWD: return launch_spmm<LPR, MMSSL_EPI_AXPY_SOFTMAX_BWD>(p, X, Y, partials, e, s);
    case MMSSL_EPI_MASK: return launch_spmm<LPR, MMSSL_EPI_MASK>(p, X, Y, partials, e, s);
  }
  return MMSSL_E_BADARG;
}

// pitched operands (column chunks of wider row-major tables): plain product and the AXPY epilogue only
template <int LPR>
int dispatch_epi_ld(const DirPlan& p, const float* X, float* Y, float* partials, int epi, const EpiArgs& e,
                    hipStream_t s) {
  switch (epi) {
    case MMSSL_EPI_NONE: return launch_spmm<LPR, MMSSL_EPI_NONE, true>(p, X, Y, partials, e, s);
    case MMSSL_EPI_AXPY: return launch_spmm<LPR, MMSSL_EPI_AXPY, true>(p, X, Y, partials, e, s);
  }
  return MMSSL_E_BADARG;
}

}  // namespace

extern "C" size_t mmssl_spmm_workspace_bytes(const mmssl_graph* g, int transpose, int d) {
  if (!g || d <= 0) return 0;
  const DirPlan& p = transpose ? g->bwd : g->fwd;
  return p.n_slots > 0 ? ws_total_bytes(p, d) : 0;
}

static int spmm_impl(const mmssl_graph* g, int transpose, const float* X, int d, float* Y, int epilogue, const float* Z,
                     float alpha, const float* S, const uint8_t* keep, int dm, void* workspace, size_t workspace_bytes,
                     void* stream);

extern "C" int mmssl_spmm_ex_f32(const mmssl_graph* g, int transpose, const float* X, int d, float* Y,
                                 int epilogue, const float* Z, float alpha, const float* S, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (epilogue == MMSSL_EPI_MASK) return MMSSL_E_BADARG;          // has its own entry point (needs the mask geometry)
  return spmm_impl(g, transpose, X, d, Y, epilogue, Z, alpha, S, nullptr, 0, workspace, workspace_bytes, stream);
}

extern "C" int mmssl_spmm_mask_f32(const mmssl_graph* g, int transpose, const float* X, int d, float* Y,
                                   const uint8_t* keep, int dm, float scale, void* workspace, size_t workspace_bytes,
                                   void* stream) {
  if (!keep || dm < 4 || (dm & 3) || d % dm != 0 || ((uintptr_t)keep & 3)) return MMSSL_E_BADARG;
  return spmm_impl(g, transpose, X, d, Y, MMSSL_EPI_MASK, nullptr, scale, nullptr, keep, dm, workspace, workspace_bytes,
                   stream);
}

static int spmm_impl(const mmssl_graph* g, int transpose, const float* X, int d, float* Y, int epilogue, const float* Z,
                     float alpha, const float* S, const uint8_t* keep, int dm, void* workspace, size_t workspace_bytes,
                     void* stream) {
  if (!g || !Y) return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  const DirPlan& p = transpose ? g->bwd : g->fwd;
  if (p.rows == 0) return 0;
  if (!X && p.nnz > 0) return MMSSL_E_BADARG;
  if ((epilogue == MMSSL_EPI_AXPY || epilogue == MMSSL_EPI_AXPY_SOFTMAX_BWD) && !Z) return MMSSL_E_BADARG;
  if (epilogue == MMSSL_EPI_AXPY_SOFTMAX_BWD && !S) return MMSSL_E_BADARG;
  if (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)workspace | (uintptr_t)Z | (uintptr_t)S) & 15) return MMSSL_E_BADARG;
  const size_t need = p.n_slots > 0 ? ws_total_bytes(p, d) : 0;
  if (need > 0 && (!workspace || workspace_bytes < need)) return MMSSL_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  float* ws = reinterpret_cast<float*>(workspace);
  EpiArgs e;
  e.Z = reinterpret_cast<const float4*>(Z);
  e.S = reinterpret_cast<const float4*>(S);
  e.alpha = alpha;
  e.keep = keep;
  e.dm = dm > 0 ? dm : d;
  e.rows = p.rows;
  e.ldx4 = e.ldy4 = d / 4;
  switch (d) {
    case 32: return dispatch_epi<8>(p, X, Y, ws, epilogue, e, s);
    case 64: return dispatch_epi<16>(p, X, Y, ws, epilogue, e, s);
    case 128: return dispatch_epi<32>(p, X, Y, ws, epilogue, e, s);
    case 256: return dispatch_epi<64>(p, X, Y, ws, epilogue, e, s);
  }
  return MMSSL_E_UNSUPP;
}

extern "C" int mmssl_spmm_ld_f32(const mmssl_graph* g, int transpose, const float* X, int64_t ldx, int d, float* Y,
                                 int64_t ldy, int epilogue, const float* Z, float alpha, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (!g || !Y) return MMSSL_E_BADARG;
  if (!supported_d(d)) return MMSSL_E_UNSUPP;
  if (epilogue != MMSSL_EPI_NONE && epilogue != MMSSL_EPI_AXPY) return MMSSL_E_BADARG;
  if (ldx < d || ldy < d || (ldx & 3) || (ldy & 3) || ldx > 0x7fffffff || ldy > 0x7fffffff) return MMSSL_E_BADARG;
  const DirPlan& p = transpose ? g->bwd : g->fwd;
  if (p.rows == 0) return 0;
  if (!X && p.nnz > 0) return MMSSL_E_BADARG;
  if (epilogue == MMSSL_EPI_AXPY && !Z) return MMSSL_E_BADARG;
  if (((uintptr_t)X | (uintptr_t)Y | (uintptr_t)workspace | (uintptr_t)Z) & 15) return MMSSL_E_BADARG;
  const size_t need = p.n_slots > 0 ? ws_total_bytes(p, d) : 0;
  if (need > 0 && (!workspace || workspace_bytes < need)) return MMSSL_E_WORKSPACE;
  hipStream_t s = as_stream(stream);
  float* ws = reinterpret_cast<float*>(workspace);
  EpiArgs e;
  e.Z = reinterpret_cast<const float4*>(Z);
  e.S = nullptr;
  e.alpha = alpha;
  e.keep = nullptr;
  e.dm = d;
  e.rows = p.rows;
  e.ldx4 = (int)(ldx / 4);
  e.ldy4 = (int)(ldy / 4);
  switch (d) {
    case 32: return dispatch_epi_ld<8>(p, X, Y, ws, epilogue, e, s);
    case 64: return dispatch_epi_ld<16>(p, X, Y, ws, epilogue, e, s);
    case 128: return dispatch_epi_ld<32>(p, X, Y, ws, epilogue, e, s);
    case 256: return dispatch_epi_ld<64>(p, X, Y, ws, epilogue, e, s);
  }
  return MMSSL_E_UNSUPP;
}

extern "C" int mmssl_spmm_f32(const mmssl_graph* g, int transpose, const float* X, int d, float* Y,
                              int epilogue, void* workspace, size_t workspace_bytes, void* stream) {
  if (epilogue != MMSSL_EPI_NONE && epilogue != MMSSL_EPI_SOFTMAX) return MMSSL_E_BADARG;
  return mmssl_spmm_ex_f32(g, transpose, X, d, Y, epilogue, nullptr, 0.f, nullptr, workspace, workspace_bytes,
                           stream);
}

// =================================================================================================
// Batch rows of the interaction PATTERN (which items a user has in the train matrix), read from the
// plan's device CSR instead of the dense `ui_graph_raw[users].todense()` the reference uploads for
// every u_sim_calculation call and for the real-data rows of the discriminator
// (/root/reference/MMSSL/main.py:281-298, 349).
//
//   mask+normalise : S[b,:] = normalize(P[b,:] * (1 - R[rows[b],:]))        (main.py:293-297)
//   its backward   : gP = (1 - R) * (gS - S (S.gS)) / |masked P|             (F.normalize backward; g/eps
//                                                                            branch for clamped rows)
//   dense rows     : out[b,:] = value * R[rows[b],:]                          (main.py:349 before the Gumbel noise)
// One block per batch row; each row is [width] floats (width = number of items, rows need not be
// 16-B aligned, so accesses are coalesced scalar loads). Deterministic (fixed-order block sums).
// =================================================================================================
namespace {

__device__ __forceinline__ float block_sum_bcast(float v, float* red) {
  v = group_sum<64>(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();                       // `red` may still be read from a previous call
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(kBlock) void rows_mask_normalize_kernel(const int32_t* __restrict__ rowptr,
                                                                     const Edge* __restrict__ edges,
                                                                     const int64_t* __restrict__ rows,
                                                                     float* __restrict__ P, int64_t width, float eps,
                                                                     float* __restrict__ inv_out) {
  __shared__ float red[4];
  const int64_t b = blockIdx.x;
  const int64_t u = rows[b];
  float* __restrict__ row = P + b * width;
  for (int e = rowptr[u] + threadIdx.x; e < rowptr[u + 1]; e += kBlock) row[edges[e].col] = 0.f;
  __syncthreads();
  float s = 0.f;
  for (int64_t j = threadIdx.x; j < width; j += kBlock) {
    const float v = row[j];
    s = fmaf(v, v, s);
  }
  const float tot = block_sum_bcast(s, red);
  const float inv = 1.f / fmaxf(sqrtf(tot), eps);
  for (int64_t j = threadIdx.x; j < width; j += kBlock) row[j] *= inv;
  if (threadIdx.x == 0) inv_out[b] = inv;
}

__global__ __launch_bounds__(kBlock) void rows_mask_normalize_bwd_kernel(const int32_t* __restrict__ rowptr,
                                                                         const Edge* __restrict__ edges,
                                                                         const int64_t* __restrict__ rows,
                                                                         const float* __restrict__ S, int64_t lds_,
                                                                         const float* __restrict__ gS, int64_t ldg,
                                                                         const float* __restrict__ inv_in,
                                                                         int64_t width, float eps,
                                                                         float* __restrict__ gP, int64_t ldp) {
  __shared__ float red[4];
  const int64_t b = blockIdx.x;
  const int64_t u = rows[b];
  const float* __restrict__ s_row = S + b * lds_;
  const float* __restrict__ g_row = gS + b * ldg;
  float* __restrict__ o_row = gP + b * ldp;
  const float inv = inv_in[b];
  float dot = 0.f;
  for (int64_t j = threadIdx.x; j < width; j += kBlock) dot = fmaf(s_row[j], g_row[j], dot);
  // inv == 1/eps exactly when the norm was clamped: F.normalize then is x/eps, gradient g/eps
  const float proj = (inv >= 1.f / eps) ? 0.f : block_sum_bcast(dot, red);
  for (int64_t j = threadIdx.x; j < width; j += kBlock) o_row[j] = inv * (g_row[j] - s_row[j] * proj);
  __syncthreads();
  for (int e = rowptr[u] + threadIdx.x; e < rowptr[u + 1]; e += kBlock) o_row[edges[e].col] = 0.f;
}

__global__ __launch_bounds__(kBlock) void rows_dense_kernel(const int32_t* __restrict__ rowptr,
                                                            const Edge* __restrict__ edges,
                                                            const int64_t* __restrict__ rows, float value,
                                                            float* __restrict__ out, int64_t width) {
  const int64_t b = blockIdx.x;
  const int64_t u = rows[b];
  float* __restrict__ row = out + b * width;
  for (int64_t j = threadIdx.x; j < width; j += kBlock) row[j] = 0.f;
  __syncthreads();
  for (int e = rowptr[u] + threadIdx.x; e < rowptr[u + 1]; e += kBlock) row[edges[e].col] = value;
}

int rows_args_ok(const mmssl_graph* g, const int64_t* rows, int64_t n, int64_t width) {
  if (!g || n < 0 || (n > 0 && !rows)) return MMSSL_E_BADARG;
  if (width != g->fwd.cols || n > 0x7fffffff) return MMSSL_E_BADARG;
  return 0;
}

}  // namespace

extern "C" int mmssl_graph_rows_mask_normalize_f32(const mmssl_graph* g, const int64_t* rows, int64_t n, float* P,
                                                   int64_t width, float eps, float* inv_norm, void* stream) {
  if (int rc = rows_args_ok(g, rows, n, width)) return rc;
  if (n == 0) return 0;
  if (!P || !inv_norm || !(eps > 0.f)) return MMSSL_E_BADARG;
  hipLaunchKernelGGL(rows_mask_normalize_kernel, dim3((unsigned)n), dim3(kBlock), 0, as_stream(stream), g->fwd.rowptr,
                     g->fwd.edges, rows, P, width, eps, inv_norm);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_graph_rows_mask_normalize_bwd_ld_f32(const mmssl_graph* g, const int64_t* rows, int64_t n,
                                                          const float* S, int64_t ld_s, const float* gS, int64_t ld_g,
                                                          const float* inv_norm, int64_t width, float eps, float* gP,
                                                          int64_t ld_p, void* stream) {
  if (int rc = rows_args_ok(g, rows, n, width)) return rc;
  if (n == 0) return 0;
  if (!S || !gS || !inv_norm || !gP || !(eps > 0.f) || ld_s < width || ld_g < width || ld_p < width) return MMSSL_E_BADARG;
  hipLaunchKernelGGL(rows_mask_normalize_bwd_kernel, dim3((unsigned)n), dim3(kBlock), 0, as_stream(stream),
                     g->fwd.rowptr, g->fwd.edges, rows, S, ld_s, gS, ld_g, inv_norm, width, eps, gP, ld_p);
  MMSSL_LAUNCH_CHECK();
  return 0;
}

extern "C" int mmssl_graph_rows_mask_normalize_bwd_f32(const mmssl_graph* g, const int64_t* rows, int64_t n,
                                                       const float* S, const float* gS, const float* inv_norm,
                                                       int64_t width, float eps, float* gP, void* stream) {
  return mmssl_graph_rows_mask_normalize_bwd_ld_f32(g, rows, n, S, width, gS, width, inv_norm, width, eps, gP, width,
                                                    stream);
}

// S[b, :] = < Q[rows[b], :], T[j, :] > with the plan's entries of row rows[b] replaced by mask_value (csrc/simtopk.hip)
extern "C" int mmssl_graph_sim_rows_f32(const mmssl_graph* g, const float* Q, const int64_t* rows, int64_t n,
                                        const float* T, int d, float mask_value, float* out, int64_t ldo,
                                        float* sumsq_part, void* stream) {
  if (!g || n < 0 || (n > 0 && (!rows || !Q || !T || !out)) || ldo < g->fwd.cols) return MMSSL_E_BADARG;
  if (n == 0 || g->fwd.cols == 0) return 0;
  if (((uintptr_t)Q | (uintptr_t)T) & 15) return MMSSL_E_BADARG;
  return sim_launch(Q, rows, n, T, g->fwd.cols, d, g->fwd.rowptr, g->fwd.edges, (int)sizeof(Edge), mask_value, nullptr, out,
                    ldo, sumsq_part, as_stream(stream));
}

// Trainer.u_sim_calculation in ONE pass over the [n, n_items] matrix: out = normalize(scores . (1 - R[rows]), dim = 1);
// the row factors come from the item table's Gram matrix before the tile kernel runs (csrc/simtopk.hip) and are returned
extern "C" int mmssl_graph_usim_rows_f32(const mmssl_graph* g, const float* Q, const int64_t* rows, int64_t n,
                                         const float* T, int d, float eps, float* out, int64_t ldo, float* inv_out,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  if (!g || n < 0 || (n > 0 && (!rows || !Q || !T || !out || !inv_out)) || ldo < g->fwd.cols || !(eps > 0.f))
    return MMSSL_E_BADARG;
  if (n == 0 || g->fwd.cols == 0) return 0;
  if (((uintptr_t)Q | (uintptr_t)T) & 15) return MMSSL_E_BADARG;
  if (d != 32 && d != 64 && d != 128) return MMSSL_E_UNSUPP;
  if (!workspace || workspace_bytes < usim_norms_workspace(d, g->fwd.cols)) return MMSSL_E_WORKSPACE;
  int rc = usim_norms_launch(Q, rows, n, T, g->fwd.cols, d, g->fwd.rowptr, g->fwd.edges, (int)sizeof(Edge), eps, inv_out,
                             workspace, as_stream(stream));
  if (rc) return rc;
  return sim_launch(Q, rows, n, T, g->fwd.cols, d, g->fwd.rowptr, g->fwd.edges, (int)sizeof(Edge), 0.f, inv_out, out, ldo,
                    nullptr, as_stream(stream));
}

extern "C" int mmssl_graph_rows_dense_f32(const mmssl_graph* g, const int64_t* rows, int64_t n, float value,
                                          float* out, int64_t width, void* stream) {
  if (int rc = rows_args_ok(g, rows, n, width)) return rc;
  if (n == 0) return 0;
  if (!out) return MMSSL_E_BADARG;
  hipLaunchKernelGGL(rows_dense_kernel, dim3((unsigned)n), dim3(kBlock), 0, as_stream(stream), g->fwd.rowptr,
                     g->fwd.edges, rows, value, out, width);
  MMSSL_LAUNCH_CHECK();
  return 0;
}
