"""GPU parity tests: every C-ABI compute entry point (through the ctypes/autograd wrappers in
mmssl_amd.ops) against the CPU oracle and the golden vectors captured from the reference.
Run on the MI355X box with `pytest -m gpu`."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import helpers as H
import mmssl_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from mmssl_amd import ops, graph
    return ops, graph


def _rand_graph(rows, cols, nnz_per_row, seed, heavy=(), empty=()):
    rng = np.random.default_rng(seed)
    r, c = [], []
    for i in range(rows):
        k = int(rng.integers(0, 2 * nnz_per_row + 1))
        r += [i] * k
        c += rng.choice(cols, size=min(k, cols), replace=False).tolist()[:k]
    for (i, k) in heavy:
        keep = [j for j in range(len(r)) if r[j] != i]
        r = [r[j] for j in keep]; c = [c[j] for j in keep]
        r += [i] * k
        c += rng.choice(cols, size=k, replace=False).tolist()
    m = sp.csr_matrix((rng.random(len(r)).astype(np.float32) + 0.1, (r, c)), shape=(rows, cols))
    m = m.tolil()
    for i in empty:
        m[i, :] = 0
    m = m.tocsr(); m.eliminate_zeros(); m.sort_indices()
    return m


@pytest.mark.parametrize("w,nc", [(64, 2), (128, 2), (128, 4), (256, 2), (256, 4)])
def test_spmm_on_column_chunks_of_wider_tables_equals_the_whole_product(w, nc):
    """mmssl_spmm_ld_f32: the product on column CHUNKS of row-major [rows, w] tables in place (row-pitched X / Y / Z; the
    sharded step's column-chunk lanes) - every plan class, both directions, plain and AXPY - against the w-wide launch:
    bit-equal on rows of at most 32 nonzeros (one lane group walks the row in edge order whatever the width), to fp32
    rounding on longer rows (their edge tiles are dealt to 64 / (width / 4) lane groups, so the summation order depends on
    the width); pitched and contiguous launches of the SAME width agree bit for bit; the stand-alone row softmax equals
    the fused epilogue bit for bit."""
    ops, graph = _ops()
    m = _rand_graph(700, 900, 6, seed=w + nc, heavy=[(5, 33), (6, 128), (7, 129), (8, 700), (699, 400), (9, 850)],
                    empty=[0, 3, 698])
    plan = graph.GraphPlan(m)
    g = torch.Generator().manual_seed(2)
    dc = w // nc
    for tr, n_in, n_out in ((False, 900, 700), (True, 700, 900)):
        X = torch.randn(n_in, w, generator=g).to(DEV)
        Z = torch.randn(n_out, w, generator=g).to(DEV)
        whole = ops._spmm_raw(plan, tr, X, ops.EPI_NONE)
        whole_axpy = ops._spmm_raw(plan, tr, X, ops.EPI_AXPY, Z, 0.25)
        Y = torch.full((n_out, w), float("nan"), device=DEV)
        Ya = torch.full((n_out, w), float("nan"), device=DEV)
        for c in range(nc):
            sl = slice(c * dc, (c + 1) * dc)
            ops._spmm_raw(plan.twin(20 + c), tr, X[:, sl], ops.EPI_NONE, out=Y[:, sl])             # pitched in, pitched out
            ops._spmm_raw(plan.twin(20 + c), tr, X[:, sl].contiguous(), ops.EPI_AXPY, Z[:, sl], 0.25, out=Ya[:, sl])
        assert not torch.isnan(Y).any() and not torch.isnan(Ya).any()          # every row of every chunk written
        assert H.rel_err(Y.cpu(), whole.cpu()) < 2e-6 and H.rel_err(Ya.cpu(), whole_axpy.cpu()) < 2e-6, (tr, w, nc)
        deg = np.diff((m.T.tocsr() if tr else m).indptr)
        short = torch.from_numpy(np.nonzero(deg <= 32)[0]).to(DEV)
        assert torch.equal(Y[short], whole[short]) and torch.equal(Ya[short], whole_axpy[short])
        # contiguous chunk out of a pitched input (what a reduce-scatter takes) == the same chunk launched all-contiguous
        P = ops._spmm_raw(plan, tr, X[:, :dc], ops.EPI_NONE)
        assert P.is_contiguous() and torch.equal(P, Y[:, :dc])
        assert torch.equal(P, ops._spmm_raw(plan, tr, X[:, :dc].contiguous(), ops.EPI_NONE))
    X = torch.randn(900, w, generator=g).to(DEV)
    fused = ops._spmm_raw(plan, False, X, ops.EPI_SOFTMAX)
    raw = ops._spmm_raw(plan, False, X, ops.EPI_NONE)
    assert torch.equal(ops.softmax_rows(raw), fused)
    assert torch.equal(ops.softmax_rows(raw, out=raw), fused)                 # in place
    y = raw.clone().requires_grad_(True)                                      # (raw is softmaxed by now: any input will do)
    sm = ops.softmax_rows_fn(y)
    gy = torch.randn(700, w, generator=g).to(DEV)
    sm.backward(gy)
    yc = y.detach().cpu().requires_grad_(True)
    torch.softmax(yc, -1).backward(gy.cpu())
    assert H.rel_err(y.grad.cpu(), yc.grad) < 1e-5
    with pytest.raises(Exception):        # the softmax epilogues need whole rows: not on a chunk
        ops._spmm_raw(plan, False, X[:, :dc], ops.EPI_SOFTMAX)


@pytest.mark.parametrize("d", [32, 64, 128])
def test_xcd_banded_plan_is_bitwise_the_flat_plan(d):
    """GraphPlan(xcd_bands=...): the short rows' work list laid out band-major (block b takes rows of column band b % 8,
    the XCD it runs on) changes which block computes which row, nothing else: forward, transpose and every epilogue equal
    the flat plan bit for bit - on a community-structured graph (banded automatically) and on a graph without locality
    (banded only when forced, unbalanced bands included)."""
    ops, graph = _ops()
    from mmssl_amd import synth
    g = torch.Generator().manual_seed(d)
    for kind, raw in (("communities", synth.interaction_matrix_communities(3000, 1900, 24000, seed=2)),
                      ("uniform", synth.interaction_matrix(3000, 1900, 24000, seed=2))):
        ui = synth.normalised_rows(raw)
        flat, auto, forced = graph.GraphPlan(ui, xcd_bands=-1), graph.GraphPlan(ui), graph.GraphPlan(ui, xcd_bands=1)
        assert not flat.info()["banded"] and forced.info()["banded"] and forced.info()["t_banded"]
        assert auto.info()["banded"] == (kind == "communities"), (kind, auto.info())
        assert (auto.info()["band_score"] > 0.8) == (kind == "communities")
        X, G = torch.randn(1900, d, generator=g).to(DEV), torch.randn(3000, d, generator=g).to(DEV)
        Z = torch.randn(3000, d, generator=g).to(DEV)
        for plan in (auto, forced):
            assert torch.equal(ops._spmm_raw(plan, False, X, ops.EPI_NONE), ops._spmm_raw(flat, False, X, ops.EPI_NONE))
            assert torch.equal(ops._spmm_raw(plan, True, G, ops.EPI_NONE), ops._spmm_raw(flat, True, G, ops.EPI_NONE))
            assert torch.equal(ops._spmm_raw(plan, False, X, ops.EPI_SOFTMAX), ops._spmm_raw(flat, False, X, ops.EPI_SOFTMAX))
            assert torch.equal(ops._spmm_raw(plan, False, X, ops.EPI_AXPY, Z, 0.5), ops._spmm_raw(flat, False, X, ops.EPI_AXPY, Z, 0.5))
            Y = torch.full((3000, d), float("nan"), device=DEV)
            ops._spmm_raw(plan, False, X[:, :d], ops.EPI_NONE, out=Y)
            assert not torch.isnan(Y).any()                  # every row written exactly once, idle band blocks included
    # communities hidden by a random renumbering of rows and columns: found by the plan-time co-clustering
    rng = np.random.default_rng(0)
    raw = synth.interaction_matrix_communities(3000, 1900, 24000, seed=2)
    ui = synth.normalised_rows(sp.csr_matrix(raw[rng.permutation(3000)][:, rng.permutation(1900)]))
    flat, auto = graph.GraphPlan(ui, xcd_bands=-1), graph.GraphPlan(ui)
    assert auto.info()["banded"] and auto.info()["t_banded"] and auto.info()["cluster_score"] > 0.6, auto.info()
    X, G = torch.randn(1900, d, generator=g).to(DEV), torch.randn(3000, d, generator=g).to(DEV)
    assert torch.equal(ops._spmm_raw(auto, False, X, ops.EPI_SOFTMAX), ops._spmm_raw(flat, False, X, ops.EPI_SOFTMAX))
    assert torch.equal(ops._spmm_raw(auto, True, G, ops.EPI_NONE), ops._spmm_raw(flat, True, G, ops.EPI_NONE))
    # empty and tiny graphs: nothing to band, nothing breaks
    e = graph.GraphPlan(sp.csr_matrix((50, 40), dtype=np.float32), xcd_bands=1)
    assert float(ops._spmm_raw(e, False, torch.randn(40, d).to(DEV), ops.EPI_NONE).abs().max()) == 0.0


@pytest.mark.parametrize("d", [32, 64, 128, 256])
def test_spmm_forward_transpose_softmax(d):
    ops, graph = _ops()
    # rows of every plan class: <=32 (lane group), 33..128 (one wave), 129..512 (one heavy block, LDS-reduced),
    # >512 (several heavy blocks -> partial slots + cross-block combine)
    m = _rand_graph(700, 900, 6, seed=d, heavy=[(5, 33), (6, 128), (7, 129), (8, 700), (699, 400), (9, 850), (10, 513)],
                    empty=[0, 3, 698])
    plan = graph.GraphPlan(m)
    info = plan.info()
    assert info["multi_rows"] >= 3 and info["wave_items"] >= 5 and info["nnz"] == m.nnz
    A = O.to_torch_sparse(m)
    g = torch.Generator().manual_seed(1)
    X = torch.randn(900, d, generator=g)
    Y = ops.spmm(plan, X.to(DEV)).cpu()
    ref = O.spmm(A, X)
    assert H.rel_err(Y, ref) < 2e-6
    assert float(Y[0].abs().max()) == 0.0 and float(Y[698].abs().max()) == 0.0   # empty rows -> exact zeros
    Ys = ops.spmm(plan, X.to(DEV), epilogue=ops.EPI_SOFTMAX).cpu()
    assert H.rel_err(Ys, torch.softmax(ref, -1)) < 1e-5      # logits reach +-30: exp() ulp differences
    np.testing.assert_allclose(Ys[0].numpy(), np.full(d, 1.0 / d, np.float32), rtol=1e-6)
    G = torch.randn(700, d, generator=g)
    Yt = ops.spmm(plan, G.to(DEV), transpose=True).cpu()
    assert H.rel_err(Yt, O.spmm(O.to_torch_sparse(m.T.tocsr()), G)) < 2e-6
    # bitwise reproducible (fixed summation order, no float atomics)
    assert torch.equal(ops.spmm(plan, X.to(DEV)).cpu(), Y)


def test_spmm_empty_graph_and_export():
    ops, graph = _ops()
    empty = sp.csr_matrix((40, 30), dtype=np.float32)
    plan = graph.GraphPlan(empty)
    assert plan._nnz() == 0
    Y = ops.spmm(plan, torch.randn(30, 64, device=DEV))
    assert Y.shape == (40, 64) and float(Y.abs().max()) == 0.0
    Yt = ops.spmm(plan, torch.randn(40, 64, device=DEV), transpose=True)
    assert Yt.shape == (30, 64) and float(Yt.abs().max()) == 0.0
    m = _rand_graph(200, 90, 5, seed=3, heavy=[(9, 80)])
    plan = graph.GraphPlan(m)
    rp, ci, va = plan.export_transpose()
    ref = m.T.tocsr(); ref.sort_indices()
    assert np.array_equal(rp, ref.indptr) and np.array_equal(ci, ref.indices) and np.array_equal(va, ref.data)


def test_spmm_autograd_matches_oracle():
    ops, graph = _ops()
    m = _rand_graph(300, 260, 8, seed=11, heavy=[(4, 200)], empty=[2])
    plan = graph.GraphPlan(m)
    A = O.to_torch_sparse(m)
    g = torch.Generator().manual_seed(2)
    X0 = torch.randn(260, 64, generator=g)
    C = torch.randn(300, 64, generator=g)
    for epi in (False, True):
        Xr = X0.clone().requires_grad_(True)
        yr = O.spmm(A, Xr)
        if epi:
            yr = torch.softmax(yr, -1)
        (yr * C).sum().backward()
        Xg = X0.clone().to(DEV).requires_grad_(True)
        yg = ops.spmm(plan, Xg, epilogue=ops.EPI_SOFTMAX if epi else ops.EPI_NONE)
        (yg * C.to(DEV)).sum().backward()
        assert H.rel_err(yg.detach().cpu(), yr.detach()) < 2e-6
        assert H.rel_err(Xg.grad.cpu(), Xr.grad) < 5e-6


def test_spmm_golden_graph():
    """The reference-normalised tiny graphs (G1) through the HIP SpMM."""
    ops, graph = _ops()
    g1 = H.load("g1_csr_norm.npz")
    for nm in ("ui", "iu"):
        shp = tuple(g1[nm + "_shape"])
        m = sp.csr_matrix((g1[nm + "_val"], (g1[nm + "_row"], g1[nm + "_col"])), shape=shp)
        X = torch.randn(shp[1], 64, generator=torch.Generator().manual_seed(5))
        Y = ops.spmm(graph.GraphPlan(m), X.to(DEV)).cpu()
        assert H.rel_err(Y, O.spmm(O.to_torch_sparse(m), X)) < 2e-6


@pytest.mark.parametrize("d", [32, 64, 128, 256])
def test_l2norm_rows_and_backward(d):
    ops, _ = _ops()
    g = torch.Generator().manual_seed(d)
    X0 = torch.randn(333, d, generator=g) * 3
    X0[7] = 0
    X0[100] = 1e-20
    base0 = torch.randn(333, d, generator=g)
    C = torch.randn(333, d, generator=g)
    for use_base, alpha in ((False, 1.0), (True, 0.36)):
        xr = X0.clone().requires_grad_(True)
        br = base0.clone().requires_grad_(True)
        yr = alpha * torch.nn.functional.normalize(xr, p=2, dim=1)
        if use_base:
            yr = br + yr
        (yr * C).sum().backward()
        xg = X0.clone().to(DEV).requires_grad_(True)
        bg = base0.clone().to(DEV).requires_grad_(True)
        yg = ops.l2norm_rows(xg, bg if use_base else None, alpha)
        (yg * C.to(DEV)).sum().backward()
        np.testing.assert_allclose(yg.detach().cpu().numpy(), yr.detach().numpy(), rtol=2e-6, atol=1e-7)
        np.testing.assert_allclose(xg.grad.cpu().numpy(), xr.grad.numpy(), rtol=2e-5, atol=2e-6)
        if use_base:
            assert torch.equal(bg.grad.cpu(), C)


def test_sumsq():
    ops, _ = _ops()
    for n in (1, 3, 4, 1023, 64 * 1000 + 2):
        x = torch.randn(n, generator=torch.Generator().manual_seed(n)).requires_grad_(True)
        xg = x.detach().to(DEV).requires_grad_(True)
        s = ops.sumsq(xg)
        ref = (x.double() ** 2).sum()
        assert abs(float(s) - float(ref)) <= 2e-6 * float(ref) + 1e-12
        (s * 0.5).backward()
        np.testing.assert_allclose(xg.grad.cpu().numpy(), x.detach().numpy(), rtol=1e-6)


@pytest.mark.parametrize("M,K,N", [(160, 32, 64), (1000, 128, 64), (777, 4096, 64), (96, 48, 64), (2049, 768, 128)])
def test_linear_forward_and_wgrad(M, K, N):
    ops, _ = _ops()
    g = torch.Generator().manual_seed(M + K)
    F_ = torch.randn(M, K, generator=g)
    W = (torch.randn(N, K, generator=g) / K ** 0.5)
    b = torch.randn(N, generator=g)
    keep = (torch.rand(M, N, generator=g) >= 0.2).to(torch.uint8)
    C = torch.randn(M, N, generator=g)
    for mask in (None, keep):
        Wr = W.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
        yr = torch.nn.functional.linear(F_, Wr, br)
        if mask is not None:
            yr = yr * mask * 1.25
        (yr * C).sum().backward()
        Wg = W.clone().to(DEV).requires_grad_(True); bg = b.clone().to(DEV).requires_grad_(True)
        yg = ops.linear(F_.to(DEV), Wg, bg, None if mask is None else mask.to(DEV), 1.25 if mask is not None else 1.0)
        (yg * C.to(DEV)).sum().backward()
        tol = 3e-6 * max(1.0, (K / 128.0) ** 0.5)
        assert H.rel_err(yg.detach().cpu(), yr.detach()) < tol
        assert H.rel_err(Wg.grad.cpu(), Wr.grad) < tol * 3
        assert H.rel_err(bg.grad.cpu(), br.grad) < tol * 3
    # asymmetric check (transpose-detecting): identity-like W picks columns of F
    Wi = torch.zeros(N, K); Wi[torch.arange(min(N, K)), torch.arange(min(N, K))] = 1.0
    y = ops.linear(F_.to(DEV), Wi.to(DEV)).cpu()
    assert torch.equal(y[:, :min(N, K)], F_[:, :min(N, K)])


def test_infonce_golden_and_oracle():
    ops, _ = _ops()
    g = H.load("g4_infonce.npz")
    tau = float(g["tau"])
    names = sorted({k.split(".")[0] for k in g.files if "." in k})
    for nm in names:
        z1 = torch.from_numpy(g[nm + ".z1"]).to(DEV).requires_grad_(True)
        z2 = torch.from_numpy(g[nm + ".z2"]).to(DEV).requires_grad_(True)
        loss = ops.infonce(z1, z2, tau)
        ref = float(g[nm + ".loss"])
        assert abs(float(loss) - ref) <= 1e-5 * abs(ref), (nm, float(loss), ref)   # north_star: 1e-4 rel
        loss.backward()
        np.testing.assert_allclose(z1.grad.cpu().numpy(), g[nm + ".gz1"], rtol=2e-4, atol=2e-7, err_msg=nm)
        np.testing.assert_allclose(z2.grad.cpu().numpy(), g[nm + ".gz2"], rtol=2e-4, atol=2e-7, err_msg=nm)
    for n, d in ((1, 64), (5, 64), (33, 32), (1024, 64), (1500, 64), (300, 256)):
        gen = torch.Generator().manual_seed(n)
        a = torch.randn(n, d, generator=gen); b = torch.randn(n, d, generator=gen)
        ar = a.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
        lr = O.infonce(ar, br, 0.5)
        (lr * 1.7).backward()
        ag = a.clone().to(DEV).requires_grad_(True); bg = b.clone().to(DEV).requires_grad_(True)
        lg = ops.infonce(ag, bg, 0.5)
        (lg * 1.7).backward()
        assert abs(float(lg) - float(lr)) <= 1e-5 * abs(float(lr)) + 3e-7, (n, d)   # n=1: log(1+1e-8) noise
        if n == 1:      # mathematically zero gradient; the reference shows ~1e-9 rounding noise
            assert float(ag.grad.abs().max()) < 1e-7 and float(bg.grad.abs().max()) < 1e-7
            continue
        assert H.rel_err(ag.grad.cpu(), ar.grad) < 2e-4, (n, d)
        assert H.rel_err(bg.grad.cpu(), br.grad) < 2e-4, (n, d)


def test_bpr_golden_and_gather():
    ops, _ = _ops()
    g = H.load("g5_bpr_featreg.npz")
    u, p, n = (torch.from_numpy(g[k]).to(DEV).requires_grad_(True) for k in ("u", "p", "n"))
    mf, emb = ops.bpr(u, p, n, float(g["decay"]), int(g["batch_size"]))
    assert abs(float(mf) - float(g["mf"])) <= 1e-6 * abs(float(g["mf"]))
    assert abs(float(emb) - float(g["emb"])) <= 1e-6 * abs(float(g["emb"]))
    (mf + emb).backward()
    for t, k in ((u, "gu"), (p, "gp"), (n, "gn")):
        np.testing.assert_allclose(t.grad.cpu().numpy(), g[k], rtol=1e-5, atol=1e-9)
    # fused-gather form vs oracle on full tables, with repeated items
    gen = torch.Generator().manual_seed(3)
    Eu = torch.randn(500, 64, generator=gen) * 0.3
    Ei = torch.randn(200, 64, generator=gen) * 0.3
    users = torch.randperm(500, generator=gen)[:128]
    pos = torch.randint(0, 200, (128,), generator=gen)
    neg = torch.randint(0, 200, (128,), generator=gen)
    Eur = Eu.clone().requires_grad_(True); Eir = Ei.clone().requires_grad_(True)
    mfr, embr, _ = O.bpr(Eur[users], Eir[pos], Eir[neg], 1e-5, 1024)
    (mfr * 0.7 + embr * 3.0).backward()
    Eug = Eu.clone().to(DEV).requires_grad_(True); Eig = Ei.clone().to(DEV).requires_grad_(True)
    mfg, embg = ops.bpr_gather(Eug, Eig, users, pos, neg, 1e-5, 1024)
    (mfg * 0.7 + embg * 3.0).backward()
    assert abs(float(mfg) - float(mfr)) <= 2e-6 * abs(float(mfr))
    assert abs(float(embg) - float(embr)) <= 2e-6 * abs(float(embr))
    assert H.rel_err(Eug.grad.cpu(), Eur.grad) < 1e-5
    assert H.rel_err(Eig.grad.cpu(), Eir.grad) < 1e-5


def test_ops_refuse_cpu_tensors():
    ops, graph = _ops()
    from mmssl_amd._lib import MmsslError
    with pytest.raises(MmsslError):
        ops.l2norm_rows(torch.randn(4, 64))
    with pytest.raises(MmsslError):
        ops.infonce(torch.randn(4, 64), torch.randn(4, 64))


def test_infonce_fused_gather_and_batch_losses():
    """idx form of InfoNCE (gather fused, scatter-add backward) and the single-node batch losses
    (BPR + 2x InfoNCE sharing one zero-filled table gradient) against the oracle."""
    ops, _ = _ops()
    gen = torch.Generator().manual_seed(21)
    U, I, B, d = 700, 300, 256, 64
    ua = torch.randn(U, d, generator=gen) * 0.5
    ia = torch.randn(I, d, generator=gen) * 0.5
    t_img = torch.randn(U, d, generator=gen)
    t_txt = torch.randn(U, d, generator=gen)
    t_txt[:50] = 0
    users = torch.randperm(U, generator=gen)[:B]
    pos = torch.randint(0, I, (B,), generator=gen)
    neg = torch.randint(0, I, (B,), generator=gen)
    w = torch.tensor([0.7, 3.0, 0.03, 0.05])
    R = [x.clone().requires_grad_(True) for x in (ua, ia, t_img, t_txt)]
    mf, emb, _ = O.bpr(R[0][users], R[1][pos], R[1][neg], 1e-5, 1024)
    c1 = O.infonce(R[2][users], R[0][users], 0.5)
    c2 = O.infonce(R[3][users], R[0][users], 0.5)
    (w[0] * mf + w[1] * emb + w[2] * c1 + w[3] * c2).backward()
    G = [x.clone().to(DEV).requires_grad_(True) for x in (ua, ia, t_img, t_txt)]
    gmf, gemb, g1, g2 = ops.batch_losses(G[0], G[1], G[2], G[3], users.to(DEV), pos.to(DEV), neg.to(DEV), 1e-5, 1024, 0.5)
    for got, ref in ((gmf, mf), (gemb, emb), (g1, c1), (g2, c2)):
        assert abs(float(got) - float(ref)) <= 1e-5 * abs(float(ref)), (float(got), float(ref))
    (w[0] * gmf + w[1] * gemb + w[2] * g1 + w[3] * g2).backward()
    for a, b in zip(G, R):
        assert H.rel_err(a.grad.cpu(), b.grad) < 2e-4
    # idx form alone == gathered form
    z1 = t_img.to(DEV).requires_grad_(True)
    z2 = ua.to(DEV).requires_grad_(True)
    l_idx = ops.infonce(z1, z2, 0.5, idx=users.to(DEV))
    l_dense = ops.infonce(t_img[users].to(DEV), ua[users].to(DEV), 0.5)
    assert float(l_idx) == float(l_dense)
    l_idx.backward()
    assert float(z1.grad[~torch.isin(torch.arange(U), users).to(DEV)].abs().max()) == 0.0


@pytest.mark.parametrize("B,d", [(300, 64), (256, 32), (33, 64)])
def test_batch_losses_eager_chain_matches_oracle_and_autograd_form(B, d):
    """The loss section as ONE chain of four launches (gradients for the promised upstream weights launched in the
    forward; InfoNCE row terms deferred to the backward pair tiles, the two losses reduced by the finish launch's guest
    block; BPR rows part in the prep launch, loss assembly + counter ticks in the finish launch:
    mmssl_infonce_multi_fwd_ticket_bpr_f32 / _bwd_phase_f32(1) / _bwd_finish_bpr_f32) against the oracle and against the
    autograd form of the same node. B = 300 and 33: ragged last tile; d = 32: two rows per LDS bank row."""
    ops, _ = _ops()
    gen = torch.Generator().manual_seed(33)
    U, I = 900, 310
    ua = torch.randn(U, d, generator=gen) * 0.5
    ia = torch.randn(I, d, generator=gen) * 0.5
    t_img = torch.randn(U, d, generator=gen)
    t_txt = torch.randn(U, d, generator=gen)
    users = torch.randperm(U, generator=gen)[:B]
    pos = torch.randint(0, I, (B,), generator=gen)          # repeated items: atomics in the scatter
    neg = torch.randint(0, I, (B,), generator=gen)
    w = torch.tensor([1.0, 1.0, 1.0, 0.03, 0.03])
    extra, c = torch.tensor(123.5), 0.25
    R = [x.clone().requires_grad_(True) for x in (ua, ia, t_img, t_txt)]
    mf, emb, _ = O.bpr(R[0][users], R[1][pos], R[1][neg], 1e-5, 1024)
    c1 = O.infonce(R[2][users], R[0][users], 0.5)
    c2 = O.infonce(R[3][users], R[0][users], 0.5)
    total_ref = w[0] * mf + w[1] * emb + w[3] * c1 + w[4] * c2 + c * extra
    total_ref.backward()
    wd = w.to(DEV)
    total = torch.zeros((), device=DEV)
    f32_tick = torch.tensor([4.0], device=DEV)
    u64_tick = torch.tensor([7], dtype=torch.int64, device=DEV)
    for rep in range(2):                                     # second round: the tickets re-armed themselves
        G = [x.clone().to(DEV).requires_grad_(True) for x in (ua, ia, t_img, t_txt)]
        terms = ops.batch_losses_vec(G[0], G[1], G[2], G[3], users.to(DEV), pos.to(DEV), neg.to(DEV), 1e-5, 1024, 0.5,
                                     eager_w=wd, tail=(extra.to(DEV), c, total, ([f32_tick.data_ptr()], [u64_tick.data_ptr()])))
        for k, ref in ((0, mf), (1, emb), (3, c1), (4, c2)):
            assert abs(float(terms[k]) - float(ref)) <= 1e-5 * abs(float(ref)), (k, float(terms[k]), float(ref))
        assert float(terms[2]) == 0.0
        assert abs(float(total) - float(total_ref)) <= 1e-5 * abs(float(total_ref))
        terms.backward(wd)
        for a, b in zip(G, R):
            assert H.rel_err(a.grad.cpu(), b.grad) < 2e-4
        assert float(f32_tick) == 5.0 + rep and int(u64_tick) == 8 + rep
    # the autograd form of the same node (backward launched by autograd, separate loss kernels): BPR terms bit for bit,
    # InfoNCE losses up to the order of the final sum (per 256-row block there, per 32-row tile here), same gradients
    # up to the order of the scatter atomics
    A = [x.clone().to(DEV).requires_grad_(True) for x in (ua, ia, t_img, t_txt)]
    t2 = ops.batch_losses_vec(A[0], A[1], A[2], A[3], users.to(DEV), pos.to(DEV), neg.to(DEV), 1e-5, 1024, 0.5)
    assert torch.equal(t2[:3], terms.detach()[:3])
    assert torch.allclose(t2[3:], terms.detach()[3:], rtol=1e-6, atol=0)
    t2.backward(wd)
    for a, b in zip(A, G):
        assert H.rel_err(a.grad.cpu(), b.grad.cpu()) < 1e-5


@pytest.mark.parametrize("G,overlap", [(1, True), (2, False), (3, True)])
def test_hot_node_matches_oracle(G, overlap):
    """The packed forward node (hotnode._HotNode: grouped projection + dropout, modal SpMM chain of width 2 d, 2G GCN
    SpMMs, layer mean + modality fusion + regulariser sum; backward = epilogue-fused SpMMs with the dropout backward in
    the last one, grouped weight gradient) vs the oracle's op-by-op autograd, incl. a gradient that arrives on a modal
    feature output itself."""
    ops, graph = _ops()
    import torch.nn.functional as F
    from mmssl_amd import hotnode
    raw = _rand_graph(500, 330, 7, seed=40 + G, heavy=[(3, 200), (9, 40)], empty=[1])
    raw.data[:] = 1.0
    ui_m, iu_m = O.csr_norm(raw, True).tocsr(), O.csr_norm(raw.T, True).tocsr()
    ui, iu = graph.GraphPlan(ui_m), graph.GraphPlan(iu_m)
    A_ui, A_iu = O.to_torch_sparse(ui_m), O.to_torch_sparse(iu_m)
    gen = torch.Generator().manual_seed(G)
    U, I, d, Ks = 500, 330, 64, (96, 160)
    Fs = [torch.randn(I, k, generator=gen) for k in Ks]
    base = [torch.randn(U, d, generator=gen), torch.randn(I, d, generator=gen)] + \
           [torch.randn(d, k, generator=gen) / k ** 0.5 for k in Ks] + [torch.randn(d, generator=gen) * 0.1 for _ in Ks]
    names = ["u0", "i0", "W_img", "W_txt", "b_img", "b_txt"]
    keep = (torch.rand(2, I, d, generator=gen) >= 0.2)
    scale = 1.25
    Cu, Ci = torch.randn(U, d, generator=gen), torch.randn(I, d, generator=gen)
    Cx = torch.randn(U, d, generator=gen)
    R = [t.clone().requires_grad_(True) for t in base]
    x = [(Fs[m] @ R[2 + m].t() + R[4 + m]) * keep[m].float() * scale for m in range(2)]
    img_u = O.spmm(A_ui, x[0]); img_i = O.spmm(A_iu, img_u)
    txt_u = O.spmm(A_ui, x[1]); txt_i = O.spmm(A_iu, txt_u)
    u_ref, i_ref = O.gcn_propagate(A_ui, A_iu, R[0], R[1], G)
    u_ref = u_ref + 0.55 * F.normalize(img_u) + 0.55 * F.normalize(txt_u)
    i_ref = i_ref + 0.55 * F.normalize(img_i) + 0.55 * F.normalize(txt_i)
    ss_ref = (img_u ** 2).sum() + (txt_u ** 2).sum() + (img_i ** 2).sum() + (txt_i ** 2).sum()
    ((u_ref * Cu).sum() + (i_ref * Ci).sum() + 0.37 * ss_ref + (txt_u * Cx).sum()).backward()
    Gt = [t.clone().to(DEV).requires_grad_(True) for t in base]
    hot = hotnode.HotCtx(torch.device(DEV, torch.cuda.current_device()), overlap=overlap)
    u_g, i_g, ss, MI, MU = hotnode.hot_node(hot, [f.to(DEV) for f in Fs], Gt[2:4], Gt[4:6], keep.to(torch.uint8).to(DEV), 0.0,
                                            scale, Gt[0], Gt[1], ui, iu, G, 0.55)
    assert MI.shape == (I, 2 * d) and MU.shape == (U, 2 * d)
    assert H.rel_err(u_g.detach().cpu(), u_ref.detach()) < 1e-5
    assert H.rel_err(i_g.detach().cpu(), i_ref.detach()) < 1e-5
    assert H.rel_err(MI[:, :d].detach().cpu(), img_i.detach()) < 1e-5 and H.rel_err(MU[:, d:].detach().cpu(), txt_u.detach()) < 1e-5
    assert abs(float(ss) - float(ss_ref)) <= 1e-5 * float(ss_ref)
    ((u_g * Cu.to(DEV)).sum() + (i_g * Ci.to(DEV)).sum() + 0.37 * ss + (MU[:, d:] * Cx.to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    for n, a, b in zip(names, Gt, R):
        assert H.rel_err(a.grad.cpu(), b.grad) < 5e-5, n


def test_hot_node_draws_masks_and_two_contexts_do_not_interfere():
    """p_drop > 0 without given masks: the node draws them in the projection's epilogue (same bytes as
    ops.dropout_masks at that generator state) and advances the generator itself; and two HotCtx objects used
    alternately on two streams reproduce what each gives alone (no shared state between step objects)."""
    ops, graph = _ops()
    from mmssl_amd import hotnode
    raw = _rand_graph(300, 200, 5, seed=3)
    raw.data[:] = 1.0
    ui, iu = graph.GraphPlan(O.csr_norm(raw, True).tocsr()), graph.GraphPlan(O.csr_norm(raw.T, True).tocsr())
    gen = torch.Generator().manual_seed(0)
    U, I, d, Ks = 300, 200, 64, (64, 128)
    dev = torch.device(DEV, torch.cuda.current_device())
    Fs = [torch.randn(I, k, generator=gen).to(DEV) for k in Ks]
    Ws = [(torch.randn(d, k, generator=gen) / k ** 0.5).to(DEV) for k in Ks]
    bs = [torch.zeros(d, device=DEV) for _ in Ks]
    u0, i0 = torch.randn(U, d, generator=gen).to(DEV), torch.randn(I, d, generator=gen).to(DEV)

    def run(hot, keep, p):
        with torch.no_grad():
            return hotnode.hot_node(hot, Fs, Ws, bs, keep, p, 1.25, u0, i0, ui, iu, 2, 0.55)
    ops.seed_dropout(5, dev)
    want = ops.dropout_masks(2, I, d, 0.2, dev)              # counter 0 -> these bytes; counter is 1 afterwards
    ref = run(hotnode.HotCtx(dev), want.contiguous(), 0.0)
    ops.seed_dropout(5, dev)
    got = run(hotnode.HotCtx(dev), None, 0.2)                # draws with counter 0, then ticks
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    assert int(ops._rng_state(dev)[1]) == 1
    # two contexts, two streams, interleaved
    h1, h2 = hotnode.HotCtx(dev), hotnode.HotCtx(dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    k1 = want.contiguous()
    k2 = (1 - want).contiguous()
    solo1, solo2 = run(hotnode.HotCtx(dev), k1, 0.0), run(hotnode.HotCtx(dev), k2, 0.0)
    torch.cuda.synchronize()
    outs = []
    for _ in range(3):
        with torch.cuda.stream(s1):
            o1 = run(h1, k1, 0.0)
        with torch.cuda.stream(s2):
            o2 = run(h2, k2, 0.0)
        outs.append((o1, o2))
    torch.cuda.synchronize()
    for o1, o2 in outs:
        assert all(torch.equal(a, b) for a, b in zip(o1, solo1)) and all(torch.equal(a, b) for a, b in zip(o2, solo2))


@pytest.mark.parametrize("d", [64, 128])
def test_spmm_inkernel_combine_stress(d):
    """Rows that span several heavy blocks are combined by the last-arriving block inside the kernel
    (write-through partials + arrival ticket). Alternate inputs launch after launch, so a stale partial from the previous
    launch (L1/L2 not refreshed) would show up as a wrong row; also check bit-reproducibility."""
    ops, graph = _ops()
    rng = np.random.default_rng(d)
    heavy = [(int(r), int(k)) for r, k in zip(rng.choice(3000, 150, replace=False), rng.integers(140, 2500, 150))]
    m = _rand_graph(3000, 2600, 5, seed=7, heavy=heavy)
    plan = graph.GraphPlan(m)
    assert plan.info()["multi_rows"] >= 100          # rows > 512 nnz span several heavy blocks
    A = O.to_torch_sparse(m)
    gen = torch.Generator().manual_seed(0)
    Xs = [torch.randn(2600, d, generator=gen) * (k + 1) for k in range(3)]
    refs = [O.spmm(A, X) for X in Xs]
    Xg = [X.to(DEV) for X in Xs]
    first = {}
    big = torch.empty(64 << 20, device=DEV)          # background traffic: evict / disturb caches
    for it in range(24):
        k = it % 3
        if it % 5 == 0:
            big.normal_()
        Y = ops.spmm(plan, Xg[k])
        Yc = Y.cpu()
        assert H.rel_err(Yc, refs[k]) < 3e-6, (it, k)
        if k in first:
            assert torch.equal(Yc, first[k]), (it, k)
        first[k] = Yc


def test_fused_adamw_matches_torch_adamw():
    """mmssl_adamw_f32 == torch.optim.AdamW (the reference's optimiser, main.py:76-80) step for step;
    tensors without a gradient are skipped; ragged sizes exercise the scalar tail."""
    from mmssl_amd.optim import FusedAdamW
    torch.manual_seed(5)
    shapes = [(1000, 64), (17, 64), (64,), (4099,), (256, 64), (3,)]
    ref_p = [torch.nn.Parameter(torch.randn(s, device=DEV)) for s in shapes]
    got_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    frozen = got_p[4].detach().clone()
    ref = torch.optim.AdamW(ref_p, lr=5.5e-4)
    got = FusedAdamW(got_p, lr=5.5e-4)
    for step in range(6):
        for k, (a, b) in enumerate(zip(ref_p, got_p)):
            if k == 4:                       # never receives a gradient (like the reference's unused tensors)
                continue
            g = torch.randn_like(a) * (10.0 ** (step - 3))
            a.grad, b.grad = g.clone(), g.clone()
        ref.step()
        got.step()
    for k, (a, b) in enumerate(zip(ref_p, got_p)):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (k, float((a - b).abs().max()))
    assert torch.equal(got_p[4], frozen)
    sd = got.state_dict()
    assert float(sd["state"][0]["step"]) == 6.0
    assert 4 not in sd["state"]
    assert torch.allclose(sd["state"][0]["exp_avg"], ref.state_dict()["state"][0]["exp_avg"], rtol=1e-5, atol=1e-7)
    # round trip through the torch-layout checkpoint
    again = FusedAdamW([torch.nn.Parameter(p.detach().clone()) for p in got_p], lr=5.5e-4)
    again.load_state_dict(sd)
    assert float(again.state_dict()["state"][0]["step"]) == 6.0


def test_fused_adamw_in_hipgraph_advances_step():
    from mmssl_amd.optim import FusedAdamW
    p = torch.nn.Parameter(torch.ones(4099, device=DEV))
    q = torch.nn.Parameter(torch.ones(4099, device=DEV))
    p.grad = torch.full_like(p, 0.5)
    q.grad = torch.full_like(q, 0.5)
    a, b = FusedAdamW([p], lr=1e-2), torch.optim.AdamW([q], lr=1e-2)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        a.step()                              # eager step 1
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):   # capture only records
            a.step()
        for _ in range(3):                    # steps 2..4
            g.replay()
    torch.cuda.synchronize()
    for _ in range(4):
        b.step()
    assert float(a.state_dict()["state"][0]["step"]) == 4.0
    assert torch.allclose(p, q, rtol=2e-6, atol=1e-7), float((p - q).abs().max())


def test_dropout_masks_statistics_and_determinism():
    from mmssl_amd import ops
    ops.seed_dropout(123, DEV)
    a = ops.dropout_masks(2, 18357, 64, 0.2, DEV)
    b = ops.dropout_masks(2, 18357, 64, 0.2, DEV)
    ops.seed_dropout(123, DEV)
    a2 = ops.dropout_masks(2, 18357, 64, 0.2, DEV)
    torch.cuda.synchronize()
    assert a.dtype == torch.uint8 and tuple(a.shape) == (2, 18357, 64)
    assert set(torch.unique(a).tolist()) == {0, 1}
    assert torch.equal(a, a2) and not torch.equal(a, b)
    n = a.numel()
    for m in (a, b):
        keep = float(m.float().mean())
        assert abs(keep - 0.8) < 5 * (0.8 * 0.2 / n) ** 0.5 + 1e-4, keep
    assert abs(float((a[0] == a[1]).float().mean()) - (0.8 * 0.8 + 0.2 * 0.2)) < 2e-3      # masks independent
    assert abs(float((a == b).float().mean()) - 0.68) < 2e-3                                 # launches independent
    # per-column / per-row keep rates are flat (no lattice structure along either axis)
    assert float((a[0].float().mean(0) - 0.8).abs().max()) < 0.02
    z = ops.dropout_masks(1, 5, 3, 0.5, DEV)                      # n not a multiple of 4
    assert tuple(z.shape) == (1, 5, 3)
    # graph replays draw fresh masks
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        ops.dropout_masks(1, 64, 64, 0.2, DEV)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            m = ops.dropout_masks(1, 64, 64, 0.2, DEV)
        g.replay()
        torch.cuda.synchronize()
        first = m.clone()
        g.replay()
        torch.cuda.synchronize()
    assert not torch.equal(first, m)


@pytest.mark.parametrize("U,I,d,B", [(300, 517, 64, 64), (50, 33, 32, 50)])
def test_usim_matches_oracle_and_dense_rows(U, I, d, B):
    """ops.usim == Trainer.u_sim_calculation (oracle restatement of main.py:281-298) forward and backward,
    with the mask read from the plan's CSR rows; graph_rows_dense == ui_graph_raw[users].todense()."""
    from mmssl_amd import ops
    from mmssl_amd.graph import GraphPlan
    rng = np.random.RandomState(U)
    raw = sp.random(U, I, density=0.05, random_state=rng, format="csr", dtype=np.float32)
    raw.data[:] = 1.0
    raw[3] = 0                                   # a user without train items
    raw.eliminate_zeros()
    plan = GraphPlan(O.csr_norm(raw, mean_flag=True))
    users = rng.choice(U, size=B, replace=B > U).tolist()
    users[0] = 3
    gen = torch.Generator().manual_seed(I)
    uf = torch.randn(U, d, generator=gen)
    itf = torch.randn(I, d, generator=gen)
    uf[users[1]] = 0.0                           # an all-zero user row: clamped norm branch
    w = torch.randn(B, I, generator=gen)
    ur, ir = uf.clone().requires_grad_(True), itf.clone().requires_grad_(True)
    ref = O.u_sim(users, ur, ir, raw, 128)
    (ref * w).sum().backward()
    ug, ig = uf.clone().to(DEV).requires_grad_(True), itf.clone().to(DEV).requires_grad_(True)
    got = ops.usim(users, ug, ig, plan)
    (got * w.to(DEV)).sum().backward()
    assert H.rel_err(got.detach().cpu(), ref.detach()) < 2e-6
    dense = np.asarray(raw[np.asarray(users)].todense())
    assert float(got.detach().cpu()[torch.from_numpy(dense > 0)].abs().max()) == 0.0      # seen items masked exactly
    assert H.rel_err(ug.grad.cpu(), ur.grad) < 2e-5
    assert H.rel_err(ig.grad.cpu(), ir.grad) < 2e-5
    rows = ops.graph_rows_dense(plan, users, 1.0)
    assert torch.equal(rows.cpu(), torch.from_numpy(dense.astype(np.float32)))


def test_spmm_concurrent_launches_on_one_plan_do_not_share_state():
    """The three chains of the fused forward/backward run SpMMs on the SAME plan at the same time on different
    streams (GraphPlan.twin = own workspace). Rows that span several blocks are combined through partial
    slots and arrival counters in that workspace: overlapping launches must not see each other's (regression:
    the counters once lived in the plan and concurrent launches corrupted the heavy rows)."""
    from mmssl_amd import ops, synth
    from mmssl_amd.graph import GraphPlan
    U, I, E, _, _ = synth.SHAPES["baby"]
    raw = synth.interaction_matrix(U, I, E, seed=1)
    ui, iu = synth.normalised_pair(raw)
    plan = GraphPlan(iu)                                   # item rows: 30 rows beyond 512 nnz
    assert plan.info()["multi_rows"] >= 10
    d = 64
    gen = torch.Generator().manual_seed(3)
    Xs = [torch.randn(U, d, generator=gen).to(DEV) for _ in range(3)]
    with torch.no_grad():
        refs = [ops._spmm_raw(plan, False, X, ops.EPI_NONE).clone() for X in Xs]      # sequential, one stream
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream() for _ in range(3)]
        views = [plan, plan.twin(1), plan.twin(2)]
        for rnd in range(30):
            outs = []
            # a spin kernel first: all three launches are queued behind it and then start TOGETHER
            # (launched one by one from Python they would never overlap on the device)
            torch.cuda._sleep(3_000_000)
            for k, st in enumerate(streams):
                st.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(st):
                    outs.append(ops._spmm_raw(views[k], False, Xs[(k + rnd) % 3], ops.EPI_NONE))
            torch.cuda.synchronize()
            for k in range(3):
                assert torch.equal(outs[k], refs[(k + rnd) % 3]), (rnd, k)


# ---------------------------------------------------------------------------------------------------
# fused similarity rows + top-K selection (csrc/simtopk.hip): evaluation scoring / ranking and u_sim forward
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("Q,n,d,B", [(500, 4500, 64, 77), (90, 33, 32, 90), (300, 2049, 128, 64), (200, 2100, 256, 45)])
def test_sim_rows_matches_fp64_product_with_csr_mask(Q, n, d, B):
    """scores of a batch against every row of the table (several 2048-item chunks, ragged last tile, batch not a
    multiple of 32), masked entries exactly mask_value, row norms from the in-kernel partials."""
    from mmssl_amd import ops
    gen = torch.Generator().manual_seed(n)
    qt, tt = torch.randn(Q, d, generator=gen), torch.randn(n, d, generator=gen)
    idx = torch.randint(0, Q, (B,), generator=gen)
    rng = np.random.RandomState(n)
    mask = sp.random(Q, n, density=0.03, random_state=rng, format="csr", dtype=np.float32)
    mask.sort_indices()
    rp = torch.from_numpy(mask.indptr.astype(np.int32)).to(DEV)
    cols = torch.from_numpy(mask.indices.astype(np.int32)).to(DEV)
    dense = torch.from_numpy(np.asarray(mask[idx.numpy()].todense()) != 0)
    ref = qt[idx].double() @ tt.double().t()
    S, _ = ops.sim_rows(qt.to(DEV), tt.to(DEV), qidx=idx.to(DEV), mask=(rp, cols), mask_value=float("-inf"))
    S = S.cpu()
    assert torch.isinf(S[dense]).all() and (S[dense] < 0).all() and torch.isfinite(S[~dense]).all()
    assert float((S[~dense].double() - ref[~dense]).abs().max()) <= 2e-6 * float(ref.abs().max())
    Sn, inv = ops.sim_rows(qt.to(DEV), tt.to(DEV), qidx=idx.to(DEV), mask=(rp, cols), mask_value=0.0, normalize=True,
                           pitch_mult=32)
    assert Sn.stride(0) % 32 == 0 and Sn.shape == (B, n)
    refn = torch.nn.functional.normalize(ref.masked_fill(dense, 0.0), dim=1)
    assert H.rel_err(Sn.cpu(), refn) < 3e-6
    S0, _ = ops.sim_rows(qt.to(DEV), tt.to(DEV))                       # no gather, no mask: plain Q . T^T
    assert float((S0.cpu().double() - qt.double() @ tt.double().t()).abs().max()) <= 2e-6 * float(ref.abs().max())


def _heapq_topk(row, k):
    """The reference's ranking rule (batch_test.py:21-36): heapq.nlargest over an ascending-id dict."""
    import heapq
    score = {i: float(v) for i, v in enumerate(row)}
    return heapq.nlargest(k, score, key=score.get)


@pytest.mark.parametrize("n,k", [(18357, 50), (700, 50), (40, 50), (5000, 1), (36864, 64), (3000, 100), (300, 256),
                                 (90000, 100), (36865, 20)])
def test_topk_rows_is_heapq_nlargest_including_ties(n, k):
    """Incl. k > 64 (block-wide winner sort) and rows wider than one launch ranks (36864 columns: block-by-block
    winners, then one launch over the winners) - the catalogue sizes / Ks the single-launch form used to refuse."""
    from mmssl_amd import ops
    gen = torch.Generator().manual_seed(n + k)
    B = 37 if n < 50000 else 9
    X = torch.randn(B, n, generator=gen)
    X[:, ::7] = X[:, 3:4]                      # many exact ties per row (every 7th column equals column 3)
    X[1] = 0.25                                # a constant row: the first k ids win
    X[2, : n // 2] = float("-inf")             # masked half
    X[3, 5] = X[3].max() + 1.0
    # mostly -inf rows (evaluation masks a user's train items to -inf): fewer than k finite scores, so the winners' tail
    # consists of REAL -inf columns in ascending column order - never a padded slot (-1) of a column block
    X[4] = float("-inf")
    X[4, [n - 3, 7, n // 2]] = torch.tensor([1.0, 2.0, 3.0])
    X[5] = float("-inf")
    idx, val = ops.topk_rows(X.to(DEV), k, values=True)
    idx, val = idx.cpu(), val.cpu()
    ke = min(k, n)
    for b in range(B):
        want = _heapq_topk(X[b].tolist(), ke)
        assert idx[b, :ke].tolist() == want, (b, idx[b, :8].tolist(), want[:8])
        assert torch.equal(val[b, :ke], X[b, torch.tensor(want)])
        assert (idx[b, ke:] == -1).all()
    # row-pitched input (the padded similarity buffer)
    Xp = torch.zeros(B, n + 13).to(DEV)
    Xp[:, :n] = X.to(DEV)
    assert torch.equal(ops.topk_rows(Xp[:, :n], k).cpu(), idx)


def test_rows_membership_matches_sets():
    from mmssl_amd import ops
    rng = np.random.RandomState(0)
    m = sp.random(64, 500, density=0.05, random_state=rng, format="csr")
    m.sort_indices()
    rows = torch.from_numpy(rng.randint(0, 64, 40).astype(np.int64))
    cand = torch.from_numpy(rng.randint(-1, 500, (40, 20)).astype(np.int64))
    out = ops.rows_membership(torch.from_numpy(m.indptr.astype(np.int32)).to(DEV),
                              torch.from_numpy(m.indices.astype(np.int32)).to(DEV), rows.to(DEV), cand.to(DEV)).cpu()
    for b in range(40):
        have = set(m.indices[m.indptr[rows[b]]:m.indptr[rows[b] + 1]].tolist())
        assert out[b].tolist() == [1 if int(c) in have else 0 for c in cand[b]]


# ---------------------------------------------------------------------------------------------------
# f-3: graph plans built on the device from (user, item) pairs (csrc/graphdev.hip)
# ---------------------------------------------------------------------------------------------------
def _pairs_reference(users, items, U, I):
    m = sp.csr_matrix((np.ones(len(users), np.float32), (users, items)), shape=(U, I))      # duplicates summed
    return O.csr_norm(m, True).tocsr(), O.csr_norm(m.T.tocsr(), True).tocsr()


@pytest.mark.parametrize("U,I,n,hot", [(700, 500, 1024, 0), (3000, 900, 9000, 4000), (50, 40, 0, 0), (400, 300, 16384, 700)])
def test_device_graph_pair_equals_scipy_rebuild_and_its_spmm(U, I, n, hot):
    """A_ui / A_iu of one rebuild (main.py:378-405) from a pair list with duplicates and (hot > 0) one item chosen by
    `hot` users — rows far beyond the 128- and 512-edge thresholds of the plan, i.e. multi-block rows combined in the
    kernel — against scipy's csr_matrix + csr_norm; then SpMM in all four directions, the softmax epilogue, autograd."""
    from mmssl_amd import ops
    from mmssl_amd.graph import DeviceGraphPair
    rng = np.random.default_rng(n + hot)
    users = rng.integers(0, U, n)
    items = rng.integers(0, I, n)
    items[:hot] = 7
    if n:
        users[-5:] = users[0]
        items[-5:] = items[0]                       # duplicates of one pair
    pair = DeviceGraphPair(U, I, max(n, 1))
    pair.rebuild(torch.from_numpy(users).to(DEV), torch.from_numpy(items).to(DEV))
    ref_ui, ref_iu = _pairs_reference(users, items, U, I)
    for plan, ref in ((pair.ui, ref_ui), (pair.iu, ref_iu)):
        got = plan.export()
        assert abs(got - ref).max() <= 2e-6 if ref.nnz else got.nnz == 0
        got_t = plan.export(transpose=True)
        assert abs(got_t - ref.T.tocsr()).max() <= 2e-6 if ref.nnz else got_t.nnz == 0
    g = torch.Generator().manual_seed(1)
    Xi, Xu = torch.randn(I, 64, generator=g), torch.randn(U, 64, generator=g)
    for plan, ref, X in ((pair.ui, ref_ui, Xi), (pair.iu, ref_iu, Xu)):
        Y = ops.spmm(plan, X.to(DEV)).cpu()
        want = torch.from_numpy(ref @ X.numpy())
        assert H.rel_err(Y, want) < 3e-6 if ref.nnz else float(Y.abs().max()) == 0.0
        Xo = Xu if X is Xi else Xi
        Yt = ops.spmm(plan, Xo.to(DEV), transpose=True).cpu()
        want_t = torch.from_numpy(ref.T.tocsr() @ Xo.numpy())
        assert H.rel_err(Yt, want_t) < 3e-6 if ref.nnz else float(Yt.abs().max()) == 0.0
        assert torch.equal(ops.spmm(plan, X.to(DEV)).cpu(), Y)                 # bitwise reproducible
    if n:
        S = ops.spmm(pair.ui, Xi.to(DEV), epilogue=ops.EPI_SOFTMAX).cpu()
        assert H.rel_err(S, torch.softmax(torch.from_numpy(ref_ui @ Xi.numpy()), -1)) < 3e-6
        xg = Xi.clone().to(DEV).requires_grad_(True)
        w = torch.randn(U, 64, generator=g)
        (ops.spmm(pair.ui, xg) * w.to(DEV)).sum().backward()
        assert H.rel_err(xg.grad.cpu(), torch.from_numpy(ref_ui.T.tocsr() @ w.numpy())) < 3e-6
    # rebuild in place with other pairs: same buffers, new graph
    users2, items2 = rng.integers(0, U, max(n // 2, 1)), rng.integers(0, I, max(n // 2, 1))
    pair.rebuild(torch.from_numpy(users2).to(DEV), torch.from_numpy(items2).to(DEV))
    r2_ui, _ = _pairs_reference(users2, items2, U, I)
    assert abs(pair.ui.export() - r2_ui).max() <= 2e-6
    assert H.rel_err(ops.spmm(pair.ui, Xi.to(DEV)).cpu(), torch.from_numpy(r2_ui @ Xi.numpy())) < 3e-6


@pytest.mark.parametrize("tag", ["k1_T1", "k2_T1", "k2_T2"])
def test_device_modal_graph_rebuild_matches_reference_loop(tag):
    """G9 (recorded from the reference's own Trainer.train(): sims in, modal graphs out) through the DEVICE path of
    Trainer._maintain_modal_graphs: top-k kernel -> pair tensors -> DeviceGraphPair.rebuild. The graphs it leaves for
    the next batch must be the reference's (values to fp32 rounding; duplicates summed)."""
    from mmssl_amd import config
    from mmssl_amd.main import Trainer
    from mmssl_amd.graph import GraphPlan
    g = H.load("g9_modal_rebuild_%s.npz" % tag)
    d, raw, U, I = H.dataset()
    config.configure([], m_topk_rate=float(g["m_topk_rate"]), T=int(g["T"]))
    tr = object.__new__(Trainer)
    tr.n_users, tr.n_items, tr.device = U, I, torch.device(DEV)
    tr._idx_cache, tr._empty_plans = (None, None), None
    tr.image_ui_index, tr.text_ui_index = {"x": [], "y": []}, {"x": [], "y": []}
    checked = 0
    for b in range(int(g["n_batches"])):
        for nm, attr in (("img_ui", "image_ui_graph"), ("img_iu", "image_iu_graph"), ("txt_ui", "text_ui_graph"),
                         ("txt_iu", "text_iu_graph")):
            cur = getattr(tr, attr, None)
            if cur is None:
                continue                              # still the initial interaction graph
            shape = (U, I) if nm.endswith("ui") else (I, U)
            want = sp.csr_matrix((g["b%d.%s_val" % (b, nm)], (g["b%d.%s_row" % (b, nm)], g["b%d.%s_col" % (b, nm)])), shape=shape)
            if isinstance(cur, GraphPlan):            # a rebuild from nothing hands out the cached empty host plan
                assert cur.nnz == 0 and want.nnz == 0 and cur.shape == shape, (tag, b, nm)
                checked += 1
                continue
            got = cur.export()
            assert got.shape == want.shape and (abs(got - want).max() <= 2e-6 if want.nnz else got.nnz == 0), (tag, b, nm)
            assert cur.nnz == int(want.sum() > 0) * cur.nnz      # nnz (host view) is 0 exactly for empty rebuilds
            checked += 1
        users = [int(u) for u in g["b%d.users" % b]]
        tr._maintain_modal_graphs(b, users, torch.from_numpy(g["b%d.img_sim" % b]).to(DEV),
                                  torch.from_numpy(g["b%d.txt_sim" % b]).to(DEV))
    assert checked >= 8
    config.configure([])


@pytest.mark.parametrize("d", [32, 256])
def test_spmm_inkernel_combine_under_graph_replay_with_five_concurrent_launches(d):
    """The cross-block combine (write-through partials + arrival ticket, no fences) at the narrow and the wide feature
    width, with FIVE launches on one plan overlapping inside a replayed hipGraph (five streams, five workspace lanes) and
    heavy background traffic between replays: every replay must reproduce the eager single-launch result bit for bit."""
    ops, graph = _ops()
    rng = np.random.default_rng(100 + d)
    heavy = [(int(r), int(k)) for r, k in zip(rng.choice(2000, 120, replace=False), rng.integers(600, 2400, 120))]
    m = _rand_graph(2000, 2500, 4, seed=9, heavy=heavy)
    plan = graph.GraphPlan(m)
    assert plan.info()["multi_rows"] >= 100
    gen = torch.Generator().manual_seed(1)
    Xs = [torch.randn(2500, d, generator=gen).to(DEV) for _ in range(5)]
    with torch.no_grad():
        want = [ops._spmm_raw(plan, False, X, ops.EPI_NONE).clone() for X in Xs]
        ref = O.spmm(O.to_torch_sparse(m), Xs[0].cpu())
        assert H.rel_err(want[0].cpu(), ref) < 3e-6
        streams = [torch.cuda.Stream() for _ in range(5)]
        main = torch.cuda.Stream()
        main.wait_stream(torch.cuda.current_stream())
        outs = [None] * 5

        def fanout():
            for k, st in enumerate(streams):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    outs[k] = ops._spmm_raw(plan.twin(k), False, Xs[k], ops.EPI_NONE)
            for st in streams:
                main.wait_stream(st)
        with torch.cuda.stream(main):
            fanout()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
                fanout()
        torch.cuda.synchronize()
        big = torch.empty(64 << 20, device=DEV)
        for it in range(12):
            if it % 3 == 0:
                big.normal_()
            g.replay()
            torch.cuda.synchronize()
            for k in range(5):
                assert torch.equal(outs[k], want[k]), (it, k)


def test_fuse_rows_equals_dense_rows_and_bwd_partials_sum_the_squares():
    """ops.fuse_fwd_rows writes exactly the dense kernel's values at the listed rows (repeats allowed); fuse_bwd's per-block
    partials add up to |Mod|^2; loss_add_partials joins c * that sum to a loss assembled without it."""
    from mmssl_amd import ops
    gen = torch.Generator().manual_seed(3)
    U, I, d, nm = 3001, 1777, 64, 2
    us = [torch.randn(U, d, generator=gen).to(DEV) for _ in range(3)]
    its = [torch.randn(I, d, generator=gen).to(DEV) for _ in range(3)]
    MU, MI = torch.randn(U, nm * d, generator=gen).to(DEV), torch.randn(I, nm * d, generator=gen).to(DEV)
    ru = torch.randint(0, U, (257,), generator=gen).to(DEV)
    ri = torch.randint(0, I, (514,), generator=gen).to(DEV)
    ri[5] = ri[6]
    dense_u, dense_i = ops.fuse_fwd([(us, MU, None), (its, MI, None)], 1 / 3, nm, 0.55)
    rows_u, rows_i = ops.fuse_fwd_rows([(us, MU, ru), (its, MI, ri)], 1 / 3, nm, 0.55)
    assert torch.equal(rows_u[ru], dense_u[ru]) and torch.equal(rows_i[ri], dense_i[ri])
    Gu, Gi = torch.randn(U, d, generator=gen).to(DEV), torch.randn(I, d, generator=gen).to(DEV)
    nbu, nbi = ops.fuse_blocks(U, d, nm), ops.fuse_blocks(I, d, nm)
    part = torch.full((nbu + nbi,), float("nan"), device=DEV)
    a = ops.fuse_bwd([(MU, Gu, None, True), (MI, Gi, None, False)], nm, 0.55, 1 / 3, None, 2.0,
                     sumsq_part=[part[:nbu], part[nbu:]])
    b = ops.fuse_bwd([(MU, Gu, None, True), (MI, Gi, None, False)], nm, 0.55, 1 / 3, None, 2.0)
    assert torch.equal(a[0][0], b[0][0]) and torch.equal(a[1][0], b[1][0]) and torch.equal(a[0][1], b[0][1])
    want = float((MU.double() ** 2).sum() + (MI.double() ** 2).sum())
    total, ss = torch.full((), 1.5, device=DEV), torch.zeros((), device=DEV)
    ops.loss_add_partials(part, 1e-3, total, ss)
    assert abs(float(ss) - want) <= 2e-6 * want
    assert abs(float(total) - (1.5 + 1e-3 * want)) <= 1e-6 * (1.5 + 1e-3 * want)


@pytest.mark.parametrize("B,K,Ks", [(700, 50, (10, 20, 50)), (1, 20, (20,)), (2049, 64, (1, 5, 64))])
def test_eval_metrics_on_the_device_equal_the_host_formulas(B, K, Ks):
    """csrc/simtopk.hip eval_metrics_kernel against utility/batch_test._metric_sums (the reference's formulas,
    batch_test.py:38-80 + metrics.py, vectorised in numpy float64): precision / recall / ndcg / hit ratio @ Ks summed
    over the users, two batches accumulated into the same totals, users without positives, candidate lists padded
    with -1; 1e-12 relative (float64 sums in another order)."""
    from mmssl_amd import ops
    from mmssl_amd.utility import batch_test
    rng = np.random.RandomState(B + K)
    U, I = 3000, 900
    pos = sp.random(U, I, density=0.01, random_state=rng, format="csr", dtype=np.float32)
    pos.sort_indices()
    rp = torch.from_numpy(pos.indptr.astype(np.int32)).to(DEV)
    cols = torch.from_numpy(pos.indices.astype(np.int32)).to(DEV)
    acc = torch.zeros((4, 8), dtype=torch.float64, device=DEV)
    want = {k: np.zeros(len(Ks)) for k in ("precision", "recall", "ndcg", "hit_ratio")}
    ws = None
    for rep in range(2):
        users = rng.choice(U, size=B, replace=B > U)
        cand = np.stack([rng.permutation(I)[:K] for _ in range(B)]).astype(np.int64)
        cand[0, K // 2:] = -1                                   # a row shorter than K
        for b in range(0, B, 7):                                # make sure there are hits at all ranks
            mine = pos.indices[pos.indptr[users[b]]:pos.indptr[users[b] + 1]]
            if len(mine):
                cand[b, rng.randint(0, K)] = mine[0]
                cand[b] = np.where(np.arange(K) == np.argmax(cand[b] == mine[0]), cand[b], np.where(cand[b] == mine[0], -1, cand[b]))
        hits = np.zeros((B, K))
        for b in range(B):
            mine = set(pos.indices[pos.indptr[users[b]]:pos.indptr[users[b] + 1]].tolist())
            hits[b] = [1.0 if (c >= 0 and int(c) in mine) else 0.0 for c in cand[b]]
        n_pos = np.diff(pos.indptr)[users]
        s = batch_test._metric_sums(hits, n_pos, list(Ks))
        for k in want:
            want[k] += s[k]
        ws = ops.eval_accumulate(rp, cols, torch.from_numpy(users.astype(np.int64)).to(DEV), torch.from_numpy(cand).to(DEV),
                                 list(Ks), acc, ws)
    got = acc.cpu().numpy()
    assert want["hit_ratio"].sum() > 0
    for m, k in enumerate(("precision", "recall", "ndcg", "hit_ratio")):
        np.testing.assert_allclose(got[m, :len(Ks)], want[k], rtol=1e-12, atol=1e-12, err_msg=k)
    assert float(np.abs(got[:, len(Ks):]).max()) == 0.0


@pytest.mark.parametrize("n,d,B", [(18357, 64, 1024), (4500, 32, 77), (2049, 128, 64)])
def test_usim_one_pass_norms_come_from_the_gram_matrix(n, d, B):
    """The one-pass u_sim (row factors from the item table's Gram matrix before the tile kernel, the [B, n] matrix written
    once, pad columns zeroed by the kernel) against float64: scores, the returned factors, exact zeros at masked entries
    and in the pad columns, and bit-identical results from two calls (fixed-order Gram sums)."""
    from mmssl_amd import ops
    gen = torch.Generator().manual_seed(n + d)
    Q = 300
    qt, tt = torch.randn(Q, d, generator=gen), torch.randn(n, d, generator=gen) * 0.3 + 0.1      # a common component
    idx = torch.randint(0, Q, (B,), generator=gen)
    rng = np.random.RandomState(n)
    mask = sp.random(Q, n, density=0.01, random_state=rng, format="csr", dtype=np.float32)
    mask.sort_indices()
    rp = torch.from_numpy(mask.indptr.astype(np.int32)).to(DEV)
    cols = torch.from_numpy(mask.indices.astype(np.int32)).to(DEV)
    dense = torch.from_numpy(np.asarray(mask[idx.numpy()].todense()) != 0)
    ref = (qt[idx].double() @ tt.double().t()).masked_fill(dense, 0.0)
    nrm = ref.norm(dim=1)
    Sn, inv = ops.sim_rows(qt.to(DEV), tt.to(DEV), qidx=idx.to(DEV), mask=(rp, cols), mask_value=0.0, normalize=True,
                           pitch_mult=32)
    full = Sn if Sn.stride(0) == Sn.shape[1] else torch.as_strided(Sn, (B, Sn.stride(0)), (Sn.stride(0), 1))
    assert float(full[:, n:].abs().max()) == 0.0 if full.shape[1] > n else True
    assert H.rel_err(Sn.cpu(), ref / nrm[:, None]) < 3e-6
    assert float(((inv.cpu().double() * nrm) - 1.0).abs().max()) < 2e-6
    assert float(Sn.cpu()[dense].abs().max()) == 0.0
    Sn2, inv2 = ops.sim_rows(qt.to(DEV), tt.to(DEV), qidx=idx.to(DEV), mask=(rp, cols), mask_value=0.0, normalize=True,
                             pitch_mult=32)
    assert torch.equal(Sn, Sn2) and torch.equal(inv, inv2)
