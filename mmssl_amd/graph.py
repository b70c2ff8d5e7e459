"""GraphPlan: the device-resident graph handle that replaces the reference's torch sparse
COO tensors (reference MMSSL/main.py:105-112,513-520: scipy -> torch.sparse.FloatTensor).

A plan owns (through libmmssl_hip.so) the CSR, its transpose (for autograd's A^T.gradY) and
a degree-balanced work list; see include/mmssl_hip.h and mmssl_amd/csrc/graph.hip.
"""
import ctypes

import numpy as np
import scipy.sparse as sp
import torch

from . import _lib


N_BANDS = 8                 # XCDs of an MI355X
CLUSTER_SCORE = 0.6         # co-clustering is used when >= this fraction of the edges stays inside its row's cluster
CLUSTER_AUTO_NNZ = 20_000_000   # automatic mode tries it up to this many entries (~0.4 s per million on the host); 2 = always


def _balanced_argmax(C, cap, rng):
    """Label per row of the score matrix C [n, k]: the best-scoring label that still has room (at most `cap` rows per
    label), the most confident rows first."""
    n, k = C.shape
    lab = np.full(n, -1, np.int64)
    room = np.full(k, cap, np.int64)
    S = C.astype(np.float64) + rng.random(C.shape) * 1e-3
    todo = np.arange(n)
    for _ in range(k):
        if todo.size == 0:
            break
        Sm = S[todo].copy()
        Sm[:, room <= 0] = -1e18
        best = Sm.argmax(1)
        conf = Sm[np.arange(todo.size), best]
        keep = np.zeros(todo.size, bool)
        for l in range(k):
            idx = np.nonzero(best == l)[0]
            if idx.size > room[l]:
                idx = idx[np.argsort(-conf[idx], kind="stable")[:room[l]]]
            keep[idx] = True
            room[l] -= idx.size
        lab[todo[keep]] = best[keep]
        todo = todo[~keep]
    if todo.size:
        lab[todo] = rng.integers(0, k, todo.size)
    return lab


def cocluster(csr, k=N_BANDS, iters=15, seed=0):
    """Balanced label propagation on the bipartite graph of a sparse matrix: rows and columns get one of k labels each so that
    a row's entries mostly lie in columns of its own label (rows take the majority label of their columns, columns of their
    rows, at most 1.08 n / k per label; random start). Returns (row_label int32 [rows], col_label int32 [cols], score) with
    score = the fraction of the entries whose row and column labels agree: ~0.4 on a graph without community structure,
    ~0.85 on one with 8 communities and 10 % global edges whatever its numbering (tests). Host-side planning for the
    XCD-banded SpMM work list (mmssl_graph_create_banded): a band need not be a contiguous range of the table - an XCD's
    L2 caches whatever rows its blocks gather."""
    rng = np.random.default_rng(seed)
    B = sp.csr_matrix((np.ones(csr.nnz, np.float32), csr.indices, csr.indptr), shape=csr.shape)
    BT = B.T.tocsr()
    R, C = B.shape
    cl = rng.integers(0, k, C)
    cap_r, cap_c = int(np.ceil(1.08 * R / k)), int(np.ceil(1.08 * C / k))
    rl = np.zeros(R, np.int64)
    coo = B.tocoo()
    score = 0.0
    for it in range(iters):
        L = np.zeros((C, k), np.float32)
        L[np.arange(C), cl] = 1.0
        rl = _balanced_argmax(B @ L, cap_r, rng)
        L = np.zeros((R, k), np.float32)
        L[np.arange(R), rl] = 1.0
        cl = _balanced_argmax(BT @ L, cap_c, rng)
        if it == 4 or it == iters - 1:
            score = float((rl[coo.row] == cl[coo.col]).mean()) if B.nnz else 0.0
            if it == 4 and score < 0.45:          # no community structure in sight (uniform graphs sit at ~0.39 here,
                break                             # graphs with communities above 0.6): do not spend the other iterations
    return rl.astype(np.int32), cl.astype(np.int32), score


class GraphPlan:
    """Sparse matrix A [rows, cols] (fp32 values) prepared for Y = A @ X on the GPU.

    Build from a scipy sparse matrix (what Trainer.csr_norm returns) or from a torch sparse
    COO tensor (the reference's handle type). The current CUDA device owns the plan.
    """

    def __init__(self, mat, device=None, xcd_bands=0):
        """xcd_bands: 0 = the work list is XCD-banded when the graph has community structure - either contiguous in the
        numbering (>= 50 % of a direction's edges in their row's dominant column band, balanced bands) or found by
        co-clustering rows and columns at plan time (`cocluster`: >= 60 % of the edges inside their row's cluster; only tried
        when the contiguous bands found nothing); 1 = always contiguous bands; 2 = always co-cluster; -1 = never. Results are
        bit-identical either way, only the block -> row assignment changes (mmssl_graph_create_ex / _banded)."""
        if isinstance(mat, torch.Tensor):
            mat = _coo_tensor_to_scipy(mat)
        csr = sp.csr_matrix(mat, dtype=np.float32)
        csr.sum_duplicates()          # torch.sparse.mm sums duplicate coordinates too
        csr.sort_indices()
        self.shape = (int(csr.shape[0]), int(csr.shape[1]))
        self.nnz = int(csr.nnz)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        rowptr = np.ascontiguousarray(csr.indptr, dtype=np.int32)
        col = np.ascontiguousarray(csr.indices, dtype=np.int32)
        val = np.ascontiguousarray(csr.data, dtype=np.float32)
        self._handle = ctypes.c_void_p()
        self._ws = {}
        rb = cb = None
        self.cluster_score = 0.0
        contiguous_tried = False
        if xcd_bands in (0, 2) and self.nnz >= 4096 and min(self.shape) >= 64 * N_BANDS:
            contiguous = False
            if xcd_bands == 0:          # contiguous column bands first (cheap): both directions must qualify
                sc = ctypes.c_double()
                band = np.empty(self.shape[0], np.int32)
                _lib.check(_lib.lib().mmssl_plan_band_host(rowptr.ctypes.data, col.ctypes.data, self.shape[0], self.shape[1],
                                                           N_BANDS, band.ctypes.data, ctypes.byref(sc)), "mmssl_plan_band_host")
                contiguous = contiguous_tried = sc.value >= 0.5
            if not contiguous and (xcd_bands == 2 or self.nnz <= CLUSTER_AUTO_NNZ):
                rl, cl, score = cocluster(csr)
                self.cluster_score = score
                if xcd_bands == 2 or score >= CLUSTER_SCORE:
                    rb, cb = np.ascontiguousarray(rl), np.ascontiguousarray(cl)
        def create(rb, cb):
            with torch.cuda.device(self.device):
                rc = _lib.lib().mmssl_graph_create_banded(
                    rowptr.ctypes.data, col.ctypes.data if self.nnz else None,
                    val.ctypes.data if self.nnz else None, self.shape[0], self.shape[1], self.nnz,
                    int(xcd_bands) if xcd_bands in (-1, 0, 1) else 0,
                    None if rb is None else rb.ctypes.data, None if cb is None else cb.ctypes.data,
                    _lib.stream_ptr(), ctypes.byref(self._handle))
            _lib.check(rc, "mmssl_graph_create_banded")
        create(rb, cb)
        if xcd_bands == 0 and contiguous_tried and rb is None and self.nnz <= CLUSTER_AUTO_NNZ:
            # the cheap pre-check saw contiguous column bands in the forward direction, but the plan builder (both directions,
            # band balance) banded NEITHER: co-clustering has not been tried yet for this graph - do so now
            i = self.info()
            if not (i["banded"] or i["t_banded"]):
                rl, cl, score = cocluster(csr)
                self.cluster_score = score
                if score >= CLUSTER_SCORE:
                    self.destroy()
                    self._handle = ctypes.c_void_p()
                    create(np.ascontiguousarray(rl), np.ascontiguousarray(cl))

    # -- reference-handle compatibility -------------------------------------------------
    def _nnz(self):
        return self.nnz

    def size(self, dim=None):
        return torch.Size(self.shape) if dim is None else self.shape[dim]

    def cuda(self, *a, **k):
        """No-op (already device resident); lets reference-style `.cuda()` chains keep working."""
        return self

    @property
    def handle(self):
        if not self._handle:
            raise _lib.MmsslError("GraphPlan used after destroy")
        return self._handle

    def info(self):
        buf = (ctypes.c_int64 * 16)()
        _lib.check(_lib.lib().mmssl_graph_info(self.handle, buf), "mmssl_graph_info")
        keys = ["rows", "cols", "nnz", "group_items", "wave_items", "multi_rows", "partial_slots", "_",
                "t_group_items", "t_wave_items", "t_multi_rows", "t_partial_slots", "short_max",
                "task_nnz", "sorted"]
        out = {k: int(v) for k, v in zip(keys, buf) if k != "_"}
        b = int(buf[15])
        out.update(banded=bool(b & 1), t_banded=bool(b & 2), band_score=((b >> 8) & 0xffff) / 1000.0,
                   t_band_score=((b >> 24) & 0xffff) / 1000.0, cluster_score=round(getattr(self, "cluster_score", 0.0), 3))
        return out

    def workspace(self, transpose, d, lane=0):
        """Scratch of one SpMM launch (partial sums of the rows that span several blocks + their arrival
        counters, which must start at zero); cached per (direction, d, lane) so pointers stay stable under
        hipGraph replay. SpMMs that may run CONCURRENTLY on different streams must use different lanes
        (see `twin`): they would otherwise share partial slots AND arrival counters."""
        key = (bool(transpose), int(d), int(lane))
        ws = self._ws.get(key)
        if ws is None:
            nbytes = _lib.lib().mmssl_spmm_workspace_bytes(self.handle, int(bool(transpose)), int(d))
            ws = torch.zeros(max(nbytes // 4, 4), dtype=torch.float32, device=self.device)
            self._ws[key] = ws
        return ws

    def twin(self, lane=1):
        """A view of this plan (same device CSR, read-only) with its own partial-sum workspace, for
        launches that overlap with launches through the plan itself on another stream."""
        return _PlanView(self, lane)

    def export_transpose(self):
        """(rowptr, col, val) numpy arrays of the device-resident transposed CSR (tests)."""
        rows, cols = self.shape
        t_rowptr = np.empty(cols + 1, np.int32)
        t_col = np.empty(max(self.nnz, 1), np.int32)
        t_val = np.empty(max(self.nnz, 1), np.float32)
        with torch.cuda.device(self.device):
            rc = _lib.lib().mmssl_graph_export_transpose(self.handle, t_rowptr.ctypes.data, t_col.ctypes.data,
                                                         t_val.ctypes.data, _lib.stream_ptr())
        _lib.check(rc, "mmssl_graph_export_transpose")
        return t_rowptr, t_col[:self.nnz], t_val[:self.nnz]

    def destroy(self):
        if self._handle:
            _lib.lib().mmssl_graph_destroy(self._handle)
            self._handle = ctypes.c_void_p()
            self._ws = {}

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class _PlanView:
    def __init__(self, plan, lane):
        self._plan, self._lane = plan, int(lane)
        self.shape, self.nnz = plan.shape, plan.nnz

    @property
    def handle(self):
        return self._plan.handle

    def workspace(self, transpose, d, lane=None):
        return self._plan.workspace(transpose, d, self._lane)

    def twin(self, lane=1):
        return _PlanView(self._plan, lane)


class _DevPlan:
    """One of the two plans of a DeviceGraphPair: same surface as GraphPlan (handle, shape, nnz, workspace, twin)."""

    def __init__(self, pair, handle, shape):
        self._pair, self._handle, self.shape = pair, handle, shape
        self.nnz = 0
        self.device = pair.device
        self._ws = {}

    @property
    def handle(self):
        if not self._pair._handle:
            raise _lib.MmsslError("DeviceGraphPair used after destroy")
        return self._handle

    def _nnz(self):
        return self.nnz

    def size(self, dim=None):
        return torch.Size(self.shape) if dim is None else self.shape[dim]

    def cuda(self, *a, **k):
        return self

    workspace = GraphPlan.workspace
    twin = GraphPlan.twin

    def export(self, transpose=False):
        """scipy CSR of A (or A^T) with duplicate pairs summed (tests; synchronises)."""
        rows = self.shape[1] if transpose else self.shape[0]
        cols = self.shape[0] if transpose else self.shape[1]
        cap = max(self._pair.capacity, 1)
        rp = np.empty(rows + 1, np.int32)
        col = np.empty(cap, np.int32)
        val = np.empty(cap, np.float32)
        nnz = ctypes.c_int64(0)
        with torch.cuda.device(self.device):
            rc = _lib.lib().mmssl_graph_export_f32(self.handle, int(bool(transpose)), rp.ctypes.data, col.ctypes.data,
                                                   val.ctypes.data, cap, ctypes.byref(nnz), _lib.stream_ptr())
        _lib.check(rc, "mmssl_graph_export_f32")
        n = int(nnz.value)
        m = sp.csr_matrix((val[:n], col[:n], rp), shape=(rows, cols))
        m.sum_duplicates()
        return m


class DeviceGraphPair:
    """Both graph plans of one modal-graph rebuild (main.py:378-405), built on the device from (user, item) pairs:
        pair = DeviceGraphPair(n_users, n_items, capacity)      # once (capacity <= 16384 pairs)
        pair.rebuild(users_idx, items_idx)                      # int64 device tensors, every T batches; no sync
        model(ui, iu, pair.ui, pair.iu, ...)
    pair.ui = csr_norm(csr_matrix(ones, (users, items)), mean_flag=True), pair.iu = csr_norm(its transpose, True)."""

    MAX_PAIRS = 16384

    def __init__(self, n_users, n_items, capacity, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.capacity = int(capacity)
        self._handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            rc = _lib.lib().mmssl_graph_pair_create(int(n_users), int(n_items), self.capacity, ctypes.byref(self._handle))
        _lib.check(rc, "mmssl_graph_pair_create")
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(_lib.lib().mmssl_graph_pair_get(self._handle, ctypes.byref(a), ctypes.byref(b)), "mmssl_graph_pair_get")
        self.ui = _DevPlan(self, a, (int(n_users), int(n_items)))
        self.iu = _DevPlan(self, b, (int(n_items), int(n_users)))

    def rebuild(self, users, items):
        users = users.to(device=self.device, dtype=torch.int64).contiguous().view(-1)
        items = items.to(device=self.device, dtype=torch.int64).contiguous().view(-1)
        n = users.shape[0]
        if items.shape[0] != n or n > self.capacity:
            raise _lib.MmsslError("DeviceGraphPair.rebuild: %d / %d pairs, capacity %d" % (n, items.shape[0], self.capacity))
        with torch.cuda.device(self.device):
            rc = _lib.lib().mmssl_graph_pair_rebuild(self._handle, users.data_ptr() if n else None,
                                                     items.data_ptr() if n else None, n, _lib.stream_ptr())
        _lib.check(rc, "mmssl_graph_pair_rebuild")
        self.ui.nnz = self.iu.nnz = n          # stored edges (duplicates are separate edges)
        return self

    def destroy(self):
        if self._handle:
            _lib.lib().mmssl_graph_pair_destroy(self._handle)
            self._handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def _coo_tensor_to_scipy(t):
    if not t.is_sparse:
        raise TypeError("expected a torch sparse COO tensor")
    t = t.detach().cpu()
    idx = t._indices().numpy()
    val = t._values().numpy().astype(np.float32)
    return sp.coo_matrix((val, (idx[0], idx[1])), shape=tuple(t.shape)).tocsr()


def as_plan(g):
    """Accept the reference's handle type (torch sparse COO), a scipy matrix or a GraphPlan."""
    return g if isinstance(g, GraphPlan) else GraphPlan(g)


# ---------------------------------------------------------------------------------------
# host-only planning helpers (no GPU needed): used by CPU tests of the host logic
# ---------------------------------------------------------------------------------------
def transpose_host(csr):
    csr = sp.csr_matrix(csr, dtype=np.float32)
    csr.sort_indices()
    rows, cols = csr.shape
    rowptr = np.ascontiguousarray(csr.indptr, np.int32)
    col = np.ascontiguousarray(csr.indices, np.int32)
    val = np.ascontiguousarray(csr.data, np.float32)
    t_rowptr = np.empty(cols + 1, np.int32)
    t_col = np.empty(max(csr.nnz, 1), np.int32)
    t_val = np.empty(max(csr.nnz, 1), np.float32)
    rc = _lib.lib().mmssl_csr_transpose_host(rowptr.ctypes.data, col.ctypes.data, val.ctypes.data, rows, cols,
                                             csr.nnz, t_rowptr.ctypes.data, t_col.ctypes.data, t_val.ctypes.data)
    _lib.check(rc, "mmssl_csr_transpose_host")
    return t_rowptr, t_col[:csr.nnz], t_val[:csr.nnz]


def plan_host(rowptr):
    rowptr = np.ascontiguousarray(rowptr, np.int32)
    rows = rowptr.shape[0] - 1
    cnt = (ctypes.c_int64 * 4)()
    _lib.check(_lib.lib().mmssl_plan_count_host(rowptr.ctypes.data, rows, cnt), "mmssl_plan_count_host")
    g = np.zeros((max(cnt[0], 1), 4), np.int32)
    w = np.zeros((max(cnt[1], 1), 4), np.int32)
    m = np.zeros((max(cnt[2], 1), 4), np.int32)
    _lib.check(_lib.lib().mmssl_plan_fill_host(rowptr.ctypes.data, rows, g.ctypes.data, w.ctypes.data,
                                               m.ctypes.data), "mmssl_plan_fill_host")
    return g[:cnt[0]], w[:cnt[1]], m[:cnt[2]], int(cnt[3])
