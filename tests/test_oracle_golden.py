"""Pins oracle/mmssl_oracle.py to the golden vectors captured from the upstream reference
(tests/golden/*.npz, produced by oracle/gen_golden.py). CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import helpers as H
import mmssl_oracle as O

TIGHT = 2e-6


def _cfg(fx=None, **kw):
    c = dict(drop_rate=0.0, batch_size=48)
    if fx is not None:
        c.update(layers=int(fx["layers"]), n_ui_layers=len(fx["weight_size"]))
    c.update(kw)
    return O.Cfg(**c)


def _graphs(fx, raw, U, I):
    ui, iu = O.graph_pair(raw)
    a, b = O.graph_pair(H.modal_raw(fx, "img", U, I))
    c, d = O.graph_pair(H.modal_raw(fx, "txt", U, I))
    return (ui, iu, a, b, c, d)


def test_g1_csr_norm():
    g = H.load("g1_csr_norm.npz")
    for src, dst, flag in (("raw", "ui", True), ("cm", "cm_norm", True), ("cm", "cm_sym", False)):
        shp = tuple(g[src + "_shape"])
        m = sp.csr_matrix((g[src + "_val"], (g[src + "_row"], g[src + "_col"])), shape=shp)
        out = O.csr_norm(m, mean_flag=flag).tocoo()
        assert np.array_equal(out.row, g[dst + "_row"]) and np.array_equal(out.col, g[dst + "_col"])
        np.testing.assert_allclose(out.data.astype(np.float32), g[dst + "_val"], rtol=1e-7, atol=0)
    shp = tuple(g["raw_shape"])
    m = sp.csr_matrix((g["raw_val"], (g["raw_row"], g["raw_col"])), shape=shp)
    out = O.csr_norm(m.T, mean_flag=True).tocoo()
    assert np.array_equal(out.row, g["iu_row"]) and np.array_equal(out.col, g["iu_col"])
    np.testing.assert_allclose(out.data.astype(np.float32), g["iu_val"], rtol=1e-7)
    # the empty rows of `cm` stay empty
    assert 4 not in set(g["cm_norm_row"]) and 9 not in set(g["cm_norm_row"])


@pytest.mark.parametrize("tag", ["g2_l1", "g3_l2"])
@pytest.mark.parametrize("modal", ["full", "sparse", "empty"])
def test_g2_forward(tag, modal):
    fx = H.load("g2_forward_%s_%s.npz" % (tag, modal))
    d, raw, U, I = H.dataset()
    P = H.params(fx)
    outs = O.forward(P, torch.from_numpy(d["image_feat"]), torch.from_numpy(d["text_feat"]),
                     _graphs(fx, raw, U, I), _cfg(fx), training=False)
    assert bool(fx["same_0_6"]) and outs[0] is outs[6] and outs[1] is outs[7]
    for n, o in zip(H.OUT_NAMES, outs):
        if n in ("ua2", "ia2"):
            continue
        np.testing.assert_allclose(o.numpy(), fx["o." + n], rtol=0, atol=TIGHT, err_msg=n)
    if modal == "empty":      # empty modal graphs -> all-zero id views
        assert float(outs[8].abs().max()) == 0.0 and float(outs[10].abs().max()) == 0.0


@pytest.mark.parametrize("tag", ["g2_l1", "g3_l2"])
@pytest.mark.parametrize("modal", ["full", "sparse"])
def test_g3_backward(tag, modal):
    fx = H.load("g3_backward_%s_%s.npz" % (tag, modal))
    d, raw, U, I = H.dataset()
    P = H.params(fx, requires_grad=True)
    outs = O.forward(P, torch.from_numpy(d["image_feat"]), torch.from_numpy(d["text_feat"]),
                     _graphs(fx, raw, U, I), _cfg(fx, drop_rate=0.0), training=True)
    scalar = sum((o * torch.from_numpy(H.cotangent(k, tuple(o.shape)))).sum() for k, o in enumerate(outs))
    assert abs(float(scalar.detach()) - float(fx["scalar"])) <= 1e-5 * abs(float(fx["scalar"])) + 1e-4
    scalar.backward()
    seen = 0
    for k in fx.files:
        if not k.startswith("g."):
            continue
        got = P[k[2:]].grad
        assert got is not None, k
        assert H.rel_err(got.numpy(), fx[k]) < 1e-5, k
        seen += 1
    assert seen >= 7
    assert P["weight_dict.w_k"].grad is None      # w_k never participates (Models.py:150)


def test_g4_infonce():
    g = H.load("g4_infonce.npz")
    names = sorted({k.split(".")[0] for k in g.files if "." in k})
    assert {"n64", "n1100_d32", "zero_z1", "some_zero_rows", "n130_d128"} <= set(names)
    for nm in names:
        z1 = torch.from_numpy(g[nm + ".z1"]).requires_grad_(True)
        z2 = torch.from_numpy(g[nm + ".z2"]).requires_grad_(True)
        loss = O.infonce(z1, z2, float(g["tau"]))
        assert abs(float(loss) - float(g[nm + ".loss"])) <= 1e-6 * abs(float(g[nm + ".loss"])), nm
        loss.backward()
        # zero rows give g/eps (1e12-scale) gradients through F.normalize: relative tolerance
        np.testing.assert_allclose(z1.grad.numpy(), g[nm + ".gz1"], rtol=2e-6, atol=1e-7, err_msg=nm)
        np.testing.assert_allclose(z2.grad.numpy(), g[nm + ".gz2"], rtol=2e-6, atol=1e-7, err_msg=nm)
    # all-zero z1 => loss = -log(1/(2N-1) + 1e-8)
    n = g["zero_z1.z1"].shape[0]
    assert abs(float(g["zero_z1.loss"]) + np.log(1.0 / (2 * n - 1) + 1e-8)) < 1e-5


def test_g5_bpr_featreg():
    g = H.load("g5_bpr_featreg.npz")
    u, p, n = (torch.from_numpy(g[k]).requires_grad_(True) for k in ("u", "p", "n"))
    mf, emb, reg = O.bpr(u, p, n, float(g["decay"]), int(g["batch_size"]))
    assert reg == 0.0
    assert abs(float(mf) - float(g["mf"])) < 1e-6 and abs(float(emb) - float(g["emb"])) < 1e-9
    (mf + emb).backward()
    for t, k in ((u, "gu"), (p, "gp"), (n, "gn")):
        np.testing.assert_allclose(t.grad.numpy(), g[k], rtol=0, atol=1e-8)
    a, b, c, d = (torch.from_numpy(g[k]).requires_grad_(True) for k in ("fa", "fb", "fc", "fd"))
    fr = O.feat_reg(a, b, c, d, int(g["n_items"]), float(g["feat_reg_decay"]))
    assert abs(float(fr) - float(g["feat_reg"])) <= 1e-6 * float(g["feat_reg"])
    fr.backward()
    np.testing.assert_allclose(a.grad.numpy(), g["gfa"], rtol=1e-6)
    np.testing.assert_allclose(c.grad.numpy(), g["gfc"], rtol=1e-6)


def test_g6_sampler_bit_exact():
    d, raw, U, I = H.dataset()
    train = H.split_lists(d, "train")
    exist = [int(u) for u, ln in zip(d["train_uid"], d["train_len"]) if ln > 0]
    train = {u: v for u, v in train.items() if len(v) > 0}
    g = H.load("g6_sample.npz")
    O.set_seed(int(g["seed"]))
    for b in range(g["users"].shape[0]):
        us, ps, ns = O.sample_bpr(exist, train, I, U, int(g["batch_size"]))
        assert np.array_equal(np.array(us), g["users"][b])
        assert np.array_equal(np.array(ps), g["pos"][b])
        assert np.array_equal(np.array(ns, np.int64), g["neg"][b])
    g = H.load("g6_sample_big.npz")
    O.set_seed(int(g["seed"]))
    us, ps, ns = O.sample_bpr(exist, train, I, U, int(g["batch_size"]))
    assert np.array_equal(np.array(us), g["users"]) and np.array_equal(np.array(ps), g["pos"])
    assert np.array_equal(np.array(ns, np.int64), g["neg"])


def test_g7_eval_recall():
    g = H.load("g7_eval.npz")
    d, raw, U, I = H.dataset()
    train = H.split_lists(d, "train")
    ua, ia = torch.from_numpy(g["ua"]), torch.from_numpy(g["ia"])
    for nm in ("val", "test"):
        pos = {u: v for u, v in H.split_lists(d, nm).items() if len(v) > 0}
        users = [int(u) for u in g[nm + ".users"]]
        assert users == list(pos.keys())
        res = O.evaluate(ua, ia, users, train, pos, [int(k) for k in g["Ks"]])
        for k in ("precision", "recall", "ndcg", "hit_ratio"):
            np.testing.assert_allclose(res[k], g["%s.%s" % (nm, k)], rtol=1e-12, atol=1e-15, err_msg=k)


@pytest.mark.parametrize("modal", ["full", "empty"])
def test_g8_gstep_assembly(modal):
    fx = H.load("g8_gstep_%s.npz" % modal)
    d, raw, U, I = H.dataset()
    P = H.params(fx, requires_grad=True)
    Dsd = {k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("D.")}
    cfg = _cfg(layers=1, n_ui_layers=2, cl_rate=float(fx["cl_rate"]), G_rate=float(fx["G_rate"]))
    users, pos, neg = (torch.from_numpy(fx[k]) for k in ("users", "pos", "neg"))
    L = O.g_step_losses(P, Dsd, torch.from_numpy(d["image_feat"]), torch.from_numpy(d["text_feat"]),
                        _graphs(fx, raw, U, I), raw, users, pos, neg, cfg)
    for k in ("mf", "emb", "feat", "cl1", "cl2", "G_lossf", "batch_loss"):
        ref = float(fx[k])
        assert abs(float(L[k]) - ref) <= 2e-6 * max(1.0, abs(ref)), (k, float(L[k]), ref)
    L["batch_loss"].backward()
    for k in fx.files:
        if k.startswith("g."):
            assert H.rel_err(P[k[2:]].grad.numpy(), fx[k]) < 2e-5, k


def test_forward_multi_with_two_modalities_is_forward_bit_for_bit():
    """The M-modality generalisation (J-1) restricted to (image, text) must be the pinned oracle forward itself."""
    fx = H.load("g2_forward_g3_l2_sparse.npz")
    d, raw, U, I = H.dataset()
    P = H.params(fx)
    cfg = O.Cfg(layers=2, n_ui_layers=3, drop_rate=0.0)
    ui, iu = O.graph_pair(raw)
    a, b = O.graph_pair(H.modal_raw(fx, "img", U, I))
    c, e = O.graph_pair(H.modal_raw(fx, "txt", U, I))
    img, txt = torch.from_numpy(d["image_feat"]), torch.from_numpy(d["text_feat"])
    ref = O.forward(P, img, txt, (ui, iu, a, b, c, e), cfg)
    got = O.forward_multi(P, [img, txt], (ui, iu), [(a, b), (c, e)], cfg)
    assert torch.equal(got["ua"], ref[0]) and torch.equal(got["ia"], ref[1])
    assert torch.equal(got["item_feats"][0], ref[2]) and torch.equal(got["item_feats"][1], ref[3])
    assert torch.equal(got["user_feats"][0], ref[4]) and torch.equal(got["user_feats"][1], ref[5])
    assert torch.equal(got["user_id"][0], ref[8]) and torch.equal(got["user_id"][1], ref[9])
    assert torch.equal(got["item_id"][0], ref[10]) and torch.equal(got["item_id"][1], ref[11])
