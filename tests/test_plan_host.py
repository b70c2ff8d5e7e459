"""CPU-only tests of the host logic inside libmmssl_hip.so (no GPU calls): the library loads,
exports every symbol include/mmssl_hip.h declares, and its CSR transpose / work-list planning
are correct."""
import os
import re

import numpy as np
import scipy.sparse as sp

from mmssl_amd import _lib, graph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rand_csr(rows, cols, density, seed, heavy_rows=()):
    rng = np.random.default_rng(seed)
    m = sp.random(rows, cols, density=density, random_state=seed, format="lil", dtype=np.float32)
    for r, k in heavy_rows:
        idx = rng.choice(cols, size=k, replace=False)
        m[r, idx] = rng.random(k).astype(np.float32) + 0.1
    m = m.tocsr()
    m.sort_indices()
    return m


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mmssl_hip.h")).read()
    declared = set(re.findall(r"\b(mmssl_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), name
    # the ctypes table binds exactly the declared set
    assert declared == set(_lib.SIGNATURES)
    assert L.mmssl_abi_version() == 1
    assert b"workspace" in L.mmssl_strerror(-3)


def test_ctypes_table_matches_the_header_prototypes():
    """Every prototype of include/mmssl_hip.h against _lib.SIGNATURES: the same number of parameters, and per parameter
    the same KIND (pointer / 64-bit integer / size_t / int / float) - a ctypes table that drifts from the header (an extra
    argument, an int where the header has int64_t) corrupts a call silently."""
    import ctypes as C
    hdr = open(os.path.join(ROOT, "include", "mmssl_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", " ", hdr, flags=re.S)
    hdr = re.sub(r"//[^\n]*", " ", hdr)
    protos = re.findall(r"\b(?:const\s+)?[A-Za-z_][A-Za-z0-9_]*\s*\*?\s*(mmssl_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)
    assert len(protos) >= 100

    def kind_of_c(param):
        t = " ".join(param.split())
        if t == "void":
            return None
        if "*" in t or "[" in t:                                  # (an array parameter is a pointer)
            return "ptr"
        base = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*$", "", t).strip()          # drop the parameter name
        base = base.replace("const ", "").strip()
        return {"int64_t": "i64", "uint64_t": "i64", "size_t": "size", "int": "int", "int32_t": "int", "uint32_t": "int",
                "float": "float", "double": "double"}[base]

    def kind_of_ct(t):
        if t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and issubclass(t, C._Pointer)):
            return "ptr"
        return {C.c_int64: "i64", C.c_uint64: "i64", C.c_size_t: "size", C.c_int: "int", C.c_int32: "int", C.c_uint32: "int",
                C.c_float: "float", C.c_double: "double"}[t]
    seen = set()
    for name, params in protos:
        want = [k for k in (kind_of_c(x) for x in params.split(",")) if k is not None]
        got = [kind_of_ct(t) for t in _lib.SIGNATURES[name][1]]
        # (size_t and 64-bit integers are the same register class; ctypes tables may use either)
        norm = lambda ks: ["i64" if k == "size" else k for k in ks]          # noqa: E731
        assert norm(got) == norm(want), (name, got, want)
        seen.add(name)
    assert seen == set(_lib.SIGNATURES)


def test_documents_name_only_declared_entry_points():
    """INTEGRATION.md / DESIGN.md / README.md / the Python layer refer to C entry points by name: every complete name they
    use must be declared in the header (INTEGRATION.md's "Removed from the ABI" paragraph is the one place where
    former names may appear)."""
    hdr = open(os.path.join(ROOT, "include", "mmssl_hip.h")).read()
    declared = set(re.findall(r"\b(mmssl_[a-z0-9_]+)\s*\(", hdr))
    suffix = r"\b(mmssl_[a-z0-9_]+_(?:f32|u8|i64|bytes|create|destroy|rebuild))\b"
    for doc in ("INTEGRATION.md", "DESIGN.md", "README.md"):
        txt = open(os.path.join(ROOT, doc)).read()
        if doc == "INTEGRATION.md":
            txt = txt.split("Removed from the ABI in round 3")[0]
        stale = sorted(n for n in set(re.findall(suffix, txt)) if n not in declared)
        assert not stale, (doc, stale)
    pkg = os.path.join(ROOT, "mmssl_amd")
    for f in sorted(os.listdir(pkg)):
        if f.endswith(".py"):
            names = set(re.findall(r"lib\(\)\.(mmssl_[a-z0-9_]+)", open(os.path.join(pkg, f)).read()))
            assert names <= declared, (f, sorted(names - declared))


def test_transpose_matches_scipy():
    for seed, (r, c, dens) in enumerate([(50, 70, 0.1), (300, 20, 0.3), (1, 9, 0.5), (40, 40, 0.0)]):
        m = _rand_csr(r, c, dens, seed)
        t_rowptr, t_col, t_val = graph.transpose_host(m)
        ref = m.T.tocsr()
        ref.sort_indices()
        assert np.array_equal(t_rowptr, ref.indptr)
        assert np.array_equal(t_col, ref.indices)      # ascending source row inside each row
        assert np.array_equal(t_val, ref.data)


def test_validate_rejects_malformed():
    L = _lib.lib()
    rowptr = np.array([0, 2, 1], np.int32)
    col = np.array([0, 1], np.int32)
    assert L.mmssl_csr_validate_host(rowptr.ctypes.data, col.ctypes.data, 2, 3, 1) == -1
    rowptr = np.array([0, 1, 2], np.int32)
    col = np.array([0, 5], np.int32)
    assert L.mmssl_csr_validate_host(rowptr.ctypes.data, col.ctypes.data, 2, 3, 2) == -1
    col = np.array([0, 2], np.int32)
    assert L.mmssl_csr_validate_host(rowptr.ctypes.data, col.ctypes.data, 2, 3, 2) == 0


def test_plan_covers_every_edge_once():
    m = _rand_csr(500, 4000, 0.004, 3, heavy_rows=[(7, 33), (8, 128), (9, 129), (10, 1000), (499, 3999)])
    m = m.tolil(); m[3, :] = 0; m = m.tocsr(); m.eliminate_zeros()
    rowptr = m.indptr.astype(np.int32)
    g, w, multi, slots = graph.plan_host(rowptr)
    deg = np.diff(rowptr)
    short_max, task = 32, 128          # defaults (MMSSL_PLAN_SHORT_MAX / MMSSL_PLAN_TASK_NNZ)
    # group items: exactly the rows with deg <= short_max, whole row each, sorted by degree desc
    assert sorted(g[:, 0].tolist()) == np.nonzero(deg <= short_max)[0].tolist()
    assert np.array_equal(g[:, 1], rowptr[g[:, 0]]) and np.array_equal(g[:, 2], rowptr[g[:, 0] + 1])
    gdeg = g[:, 2] - g[:, 1]
    assert np.all(np.diff(gdeg) <= 0)
    assert 3 in g[:, 0] and gdeg[-1] == 0
    # wave items tile the long rows; the list is [heavy section (whole blocks of 4 per row) | light section]
    covered = np.zeros(m.nnz, np.int32)
    for row, beg, end, code in w:
        if row < 0:                                  # padding of a heavy row's last block: no work
            assert beg == 0 and end == 0 and code != -1
            continue
        assert deg[row] > short_max and 0 < end - beg <= task
        covered[beg:end] += 1
    for row, beg, end, slot in g:
        covered[beg:end] += 1
    assert np.all(covered == 1)
    heavy = w[w[:, 3] != -1]
    light = w[w[:, 3] == -1]
    assert len(heavy) % 4 == 0 and np.array_equal(w[:len(heavy)], heavy)        # heavy section first
    assert all(short_max < deg[r] <= task for r in light[:, 0])
    assert sorted(light[:, 0].tolist()) == np.nonzero((deg > short_max) & (deg <= task))[0].tolist()
    # every block of the heavy section holds slices of ONE row, real slices first, one code per block
    hdeg = []
    for b0 in range(0, len(heavy), 4):
        blk = heavy[b0:b0 + 4]
        assert blk[0, 0] >= 0 and len(set(blk[:, 3].tolist())) == 1
        real = blk[blk[:, 0] >= 0]
        assert len(set(real[:, 0].tolist())) == 1 and np.all(blk[:len(real), 0] >= 0)
        assert deg[blk[0, 0]] > task
        hdeg.append(int(deg[blk[0, 0]]))
    assert all(a >= b for a, b in zip(hdeg, hdeg[1:]))                           # heaviest rows first
    blocks_of = lambda d: -(-(-(-int(d) // task)) // 4)                          # noqa: E731  ceil(ceil(d/task)/4)
    # rows that fit one block (deg <= 4*task) need no partial slot; the others own one slot per block
    assert slots == int(sum(blocks_of(deg[r]) for r in np.nonzero(deg > 4 * task)[0]))
    seen = []
    for row, first, n, _ in multi:
        assert deg[row] > 4 * task and n == blocks_of(deg[row]) and n > 1
        mine = heavy[heavy[:, 0] == row]
        assert len(mine) == -(-int(deg[row]) // task)
        assert mine[:, 3].tolist() == [first + k // 4 for k in range(len(mine))]
        seen += list(range(first, first + n))
    assert sorted(seen) == list(range(slots))
    fit = [r for r in np.nonzero((deg > task) & (deg <= 4 * task))[0]]
    for r in fit:
        assert set(heavy[heavy[:, 0] == r][:, 3].tolist()) == {-2}


def test_xcd_band_host_helpers():
    """mmssl_plan_band_host / mmssl_plan_band_group_items_host (pure CPU): dominant column band per row, locality score,
    stable band-major reordering of the degree-sorted group items."""
    import ctypes
    import scipy.sparse as sp
    from mmssl_amd import _lib, synth
    L = _lib.lib()

    def bands_of(mat, n_bands=8):
        mat = sp.csr_matrix(mat)
        mat.sort_indices()
        rp = np.ascontiguousarray(mat.indptr, dtype=np.int32)
        col = np.ascontiguousarray(mat.indices, dtype=np.int32)
        band = np.empty(mat.shape[0], np.int32)
        sc = ctypes.c_double()
        assert L.mmssl_plan_band_host(rp.ctypes.data, col.ctypes.data, mat.shape[0], mat.shape[1], n_bands,
                                      band.ctypes.data, ctypes.byref(sc)) == 0
        return rp, band, sc.value
    # a block-diagonal graph with 10 % global edges: nearly every edge in its row's band; the plain generator: no locality
    com = synth.interaction_matrix_communities(4000, 2400, 30000, n_comm=8, cross=0.1, seed=3)
    uni = synth.interaction_matrix(4000, 2400, 30000, seed=3)
    rp, band, score = bands_of(com)
    assert score > 0.85 and bands_of(uni)[2] < 0.5
    rows = np.arange(4000)
    assert (band[rows < 3500] == (rows[rows < 3500] // 500)).mean() > 0.95          # the community IS the band
    # reference restatement of the band choice (ties -> lowest band, empty rows -> row % n_bands)
    m = sp.csr_matrix(com)
    width = -(-2400 // 8)
    for r in (0, 17, 1999, 3999):
        cols = m.indices[m.indptr[r]:m.indptr[r + 1]]
        cnt = np.bincount(np.minimum(cols // width, 7), minlength=8)
        assert band[r] == (int(np.argmax(cnt)) if len(cols) else r % 8)
    # group items: degree-sorted list -> band-major, order inside a band preserved
    counts = (ctypes.c_int64 * 4)()
    assert L.mmssl_plan_count_host(rp.ctypes.data, 4000, counts) == 0
    n_g, n_w, n_m = counts[0], counts[1], counts[2]
    gi, wi, mi = (np.zeros(max(4 * n, 4), np.int32) for n in (n_g, n_w, n_m))
    assert L.mmssl_plan_fill_host(rp.ctypes.data, 4000, gi.ctypes.data, wi.ctypes.data, mi.ctypes.data) == 0
    before = gi.reshape(-1, 4)[:n_g].copy()
    start = np.zeros(9, np.int32)
    assert L.mmssl_plan_band_group_items_host(gi.ctypes.data, n_g, band.ctypes.data, 8, start.ctypes.data) == 0
    after = gi.reshape(-1, 4)[:n_g]
    assert start[0] == 0 and start[8] == n_g and (np.diff(start) >= 0).all()
    for x in range(8):
        seg = after[start[x]:start[x + 1]]
        assert (band[seg[:, 0]] == x).all()
        want = before[band[before[:, 0]] == x]
        assert np.array_equal(seg, want)                                             # stable: degree order kept


def test_xcd_band_wave_block_map_is_a_permutation_that_follows_the_bands():
    import ctypes
    import scipy.sparse as sp
    from mmssl_amd import _lib, synth
    L = _lib.lib()
    m = sp.csr_matrix(synth.interaction_matrix_communities(6000, 1600, 90000, n_comm=8, cross=0.1, seed=5).T)   # item rows: many long ones
    m.sort_indices()
    rows = m.shape[0]
    rp = np.ascontiguousarray(m.indptr, dtype=np.int32)
    col = np.ascontiguousarray(m.indices, dtype=np.int32)
    band = np.empty(rows, np.int32)
    assert L.mmssl_plan_band_host(rp.ctypes.data, col.ctypes.data, rows, m.shape[1], 8, band.ctypes.data, None) == 0
    counts = (ctypes.c_int64 * 4)()
    assert L.mmssl_plan_count_host(rp.ctypes.data, rows, counts) == 0
    n_g, n_w, n_m = counts[0], counts[1], counts[2]
    assert n_w > 200 and n_m > 0
    gi, wi, mi = (np.zeros(max(4 * n, 4), np.int32) for n in (n_g, n_w, n_m))
    assert L.mmssl_plan_fill_host(rp.ctypes.data, rows, gi.ctypes.data, wi.ctypes.data, mi.ctypes.data) == 0
    before = wi.reshape(-1, 4)[:n_w].copy()
    nb = (n_w + 3) // 4
    wmap = np.full(nb, -1, np.int32)
    assert L.mmssl_plan_band_wave_blocks_host(wi.ctypes.data, n_w, band.ctypes.data, 8, wmap.ctypes.data) == 0
    after = wi.reshape(-1, 4)[:n_w]
    assert sorted(wmap.tolist()) == list(range(nb))                                  # every wave block exactly once
    light = (before[:, 3] == -1) & (before[:, 0] >= 0)
    l0 = int(np.argmax(light))
    assert np.array_equal(before[:l0], after[:l0])                                   # heavy section untouched
    assert sorted(map(tuple, before[l0:])) == sorted(map(tuple, after[l0:]))         # light rows: same set, band-major
    assert (np.diff(band[after[l0:, 0]]) >= 0).all()
    # the block a hardware block gets belongs to its band while that band has blocks left
    blk_band = band[after[::4, 0]]
    left = np.bincount(blk_band, minlength=8)
    for b in range(nb):
        x = b % 8
        if left[x] > 0:
            assert blk_band[wmap[b]] == x
        left[blk_band[wmap[b]]] -= 1


def test_cocluster_finds_communities_in_any_numbering():
    """graph.cocluster (plan-time co-clustering for the XCD-banded work list): 8 planted communities with 10 % global edges,
    users and items randomly renumbered - the contiguous column bands see nothing, the co-clustering puts > 75 % of the edges
    inside their row's cluster with balanced clusters; a graph without communities stays below the 0.6 threshold."""
    import ctypes
    import scipy.sparse as sp
    from mmssl_amd import _lib, graph, synth
    U, I = 6000, 3200
    com = synth.interaction_matrix_communities(U, I, 48000, n_comm=8, cross=0.1, seed=7)
    rng = np.random.default_rng(1)
    perm = sp.csr_matrix(com[rng.permutation(U)][:, rng.permutation(I)])
    perm.sort_indices()
    sc = ctypes.c_double()
    band = np.empty(U, np.int32)
    rp, col = np.ascontiguousarray(perm.indptr, dtype=np.int32), np.ascontiguousarray(perm.indices, dtype=np.int32)
    assert _lib.lib().mmssl_plan_band_host(rp.ctypes.data, col.ctypes.data, U, I, 8, band.ctypes.data, ctypes.byref(sc)) == 0
    assert sc.value < 0.5                                            # contiguous bands: the numbering hides the communities
    rl, cl, score = graph.cocluster(perm)
    assert score > 0.75, score
    assert rl.dtype == np.int32 and rl.shape == (U,) and cl.shape == (I,) and rl.min() >= 0 and cl.max() < 8
    assert np.bincount(rl, minlength=8).max() <= 1.09 * U / 8 + 1 and np.bincount(cl, minlength=8).max() <= 1.09 * I / 8 + 1
    uni = sp.csr_matrix(synth.interaction_matrix(U, I, 48000, seed=7))
    assert graph.cocluster(uni)[2] < graph.CLUSTER_SCORE
