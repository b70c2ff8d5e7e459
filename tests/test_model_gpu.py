"""GPU parity of the assembled hot path (MMSSL.forward / backward, the generator-step loss,
a short Trainer run) against golden vectors captured from the reference and the CPU oracle."""
import math

import numpy as np
import pytest
import scipy.sparse as sp
import torch

import helpers as H
import mmssl_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _configure(**kw):
    from mmssl_amd import config
    base = dict(drop_rate=0.0, batch_size=48, debug=True)
    base.update(kw)
    return config.configure([], **base)


def _model(fx, layers, weight_size, feats):
    from mmssl_amd.Models import MMSSL
    _configure(layers=layers)
    m = MMSSL(feats[2], feats[3], 64, list(weight_size), [0.1] * len(weight_size), feats[0], feats[1])
    missing, unexpected = m.load_state_dict({k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("p.")},
                                            strict=False)
    assert not unexpected
    return m.to(DEV)


def _plans(fx, raw, U, I):
    from mmssl_amd.graph import GraphPlan
    from mmssl_amd import synth

    def pair(m):
        a, b = O.csr_norm(m, True), O.csr_norm(m.T, True)     # host graph prep == reference csr_norm
        return GraphPlan(a), GraphPlan(b)
    ui, iu = pair(raw)
    a, b = pair(H.modal_raw(fx, "img", U, I))
    c, d = pair(H.modal_raw(fx, "txt", U, I))
    return ui, iu, a, b, c, d


@pytest.mark.parametrize("tag,layers", [("g2_l1", 1), ("g3_l2", 2)])
@pytest.mark.parametrize("modal", ["full", "sparse", "empty"])
def test_forward_matches_reference(tag, layers, modal):
    fx = H.load("g2_forward_%s_%s.npz" % (tag, modal))
    d, raw, U, I = H.dataset()
    m = _model(fx, layers, fx["weight_size"], (d["image_feat"], d["text_feat"], U, I)).eval()
    with torch.no_grad():
        outs = m(*_plans(fx, raw, U, I))
    assert outs[0] is outs[6] and outs[1] is outs[7] and len(outs) == 12
    for n, o in zip(H.OUT_NAMES, outs):
        if n in ("ua2", "ia2"):
            continue
        ref = fx["o." + n]
        # folded modality fusion differs from the literal 5-D form by <= ~4e-6
        np.testing.assert_allclose(o.cpu().numpy(), ref, rtol=2e-5, atol=2e-5, err_msg=n)
        assert H.rel_err(o.cpu(), ref) < 1e-4 or float(np.abs(ref).max()) == 0.0, n


def test_forward_accepts_torch_sparse_graphs():
    """Reference handle type (torch sparse COO) in, same result as GraphPlan in."""
    fx = H.load("g2_forward_g2_l1_sparse.npz")
    d, raw, U, I = H.dataset()
    m = _model(fx, 1, fx["weight_size"], (d["image_feat"], d["text_feat"], U, I)).eval()
    plans = _plans(fx, raw, U, I)
    coo = [O.to_torch_sparse(O.csr_norm(x, True)) for x in
           (raw, raw.T, H.modal_raw(fx, "img", U, I), H.modal_raw(fx, "img", U, I).T,
            H.modal_raw(fx, "txt", U, I), H.modal_raw(fx, "txt", U, I).T)]
    with torch.no_grad():
        a = m(*plans)
        b = m(*coo)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("tag,layers", [("g2_l1", 1), ("g3_l2", 2)])
@pytest.mark.parametrize("modal", ["full", "sparse"])
def test_backward_matches_reference(tag, layers, modal):
    fx = H.load("g3_backward_%s_%s.npz" % (tag, modal))
    d, raw, U, I = H.dataset()
    m = _model(fx, layers, fx["weight_size"], (d["image_feat"], d["text_feat"], U, I)).train()
    outs = m(*_plans(fx, raw, U, I))
    scalar = sum((o * torch.from_numpy(H.cotangent(k, tuple(o.shape))).to(DEV)).sum() for k, o in enumerate(outs))
    assert abs(float(scalar) - float(fx["scalar"])) <= 1e-4 * abs(float(fx["scalar"])) + 1e-3
    scalar.backward()
    named = dict(m.named_parameters())
    checked = 0
    for k in fx.files:
        if not k.startswith("g."):
            continue
        name = k[2:]
        if name == "weight_dict.w_q":     # numerical-noise-only gradient in the reference (|g| ~ 1e-9)
            assert float(np.abs(fx[k]).max()) < 1e-5
            continue
        got = named[name].grad
        assert got is not None, name
        assert H.rel_err(got.cpu(), fx[k]) < 1e-4, (name, H.rel_err(got.cpu(), fx[k]))
        checked += 1
    assert checked >= 7


def test_hotpath_weight_planes_are_made_behind_the_update_and_follow_external_writes():
    """HotPathStep keeps the projection weights' bf16 planes as a step-owned image (ops.WeightPlanes) that the weight
    gradient's epilogue rewrites together with the weights (first step: a split launch behind the update): the captured
    forward starts with the projection's main kernel. (1) the trajectory equals the one of
    a step object without the image (its forward splits the weights itself), eager and replayed; (2) a write to the
    weights from outside (what load_state_dict / a torch optimiser do: an in-place op on the Parameter) between two replays
    is noticed by run() and the image remade - the next loss equals the one of an object that never had an image."""
    import scipy.sparse as sp
    from mmssl_amd import ops, synth
    from mmssl_amd.graph import GraphPlan
    from mmssl_amd.hotpath import HotPathStep
    from mmssl_amd.Models import MMSSL
    U, I, E, dv, dt, B = 2000, 1300, 20000, 260, 100, 256        # (widths that end inside a 32-deep slice: zero padding)
    _configure(drop_rate=0.2, batch_size=B, weight_size="[64, 64]")
    raw = synth.interaction_matrix(U, I, E, seed=9)
    ui, iu = synth.normalised_pair(raw)
    g = torch.Generator().manual_seed(5)
    img, txt = torch.randn(I, dv, generator=g).numpy(), torch.randn(I, dt, generator=g).numpy()
    batches = [(torch.randperm(U, generator=g)[:B], torch.randint(0, I, (B,), generator=g),
                torch.randint(0, I, (B,), generator=g)) for _ in range(6)]
    bump = torch.randn(64, dv, generator=g) * 0.01

    def run(planes, capture):
        torch.manual_seed(21)
        ops.seed_dropout(21)
        model = MMSSL(U, I, 64, [64] * 2, [0.1] * 2, img, txt).to(DEV).train()
        e1, e2 = GraphPlan(sp.csr_matrix((U, I), dtype=np.float32)), GraphPlan(sp.csr_matrix((I, U), dtype=np.float32))
        step = HotPathStep(model, (GraphPlan(ui), GraphPlan(iu), e1, e2, e1, e2), B)
        if not planes:
            step.hot.planes = None
        else:
            assert step.hot.planes is not None
        step.set_batch(*[t.to(DEV) for t in batches[0]])
        if capture:
            assert step.capture(warmup=1), getattr(step, "capture_error", "")
        else:
            step.step()
        losses = []
        for k, b in enumerate(batches):
            if k == 3:                 # somebody else writes the weights (an in-place op on the Parameter, like copy_)
                with torch.no_grad():
                    model.image_trans.weight.add_(bump.to(DEV))
            step.set_batch(*[t.to(DEV) for t in b])
            step.run()
            torch.cuda.synchronize()
            losses.append(float(step.loss))
        if planes:
            ws = [model.image_trans.weight, model.text_trans.weight]
            got = step.hot.planes.image_for(ws)
            assert got is not None
            # the image the weight-gradient epilogue has been rewriting in place, step after step, is bit for bit the split
            # of the weights as they are now (incl. the zero padding of the last slices)
            fresh = ops.WeightPlanes()
            assert fresh.refresh(ws)
            torch.cuda.synchronize()
            assert torch.equal(got.view(torch.int32), fresh.buf.view(torch.int32))
        return losses, model.image_trans.weight.detach().cpu().clone()

    ref_l, ref_w = run(False, False)
    for planes, capture in ((True, False), (True, True), (False, True)):
        l, w = run(planes, capture)
        for a, b in zip(l, ref_l):
            assert abs(a - b) <= 1e-5 * abs(b), (planes, capture, l, ref_l)
        assert H.rel_err(w, ref_w) < 1e-5, (planes, capture)


def _trainer(tmp_path, **kw):
    from mmssl_amd import config
    from mmssl_amd.utility import batch_test
    root = H.write_dataset_dir(str(tmp_path))
    _configure(data_path=root, dataset="tiny", **kw)
    batch_test.init_data()
    from mmssl_amd.main import Trainer, set_seed
    set_seed(2022)
    return Trainer(data_config={})


@pytest.mark.parametrize("modal", ["full", "empty"])
def test_generator_step_assembly_matches_reference(tmp_path, modal):
    fx = H.load("g8_gstep_%s.npz" % modal)
    d, raw, U, I = H.dataset()
    tr = _trainer(tmp_path)
    tr.model.load_state_dict({k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("p.")}, strict=False)
    tr.D.load_state_dict({k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("D.")})
    tr.D.eval()
    tr.model.train()
    ui, iu, a, b, c, dd = _plans(fx, raw, U, I)
    tr.image_ui_graph, tr.image_iu_graph, tr.text_ui_graph, tr.text_iu_graph = a, b, c, dd
    L = tr.generator_losses(0, fx["users"].tolist(), fx["pos"].tolist(), fx["neg"].tolist(), maintain_graphs=False)
    for k in ("mf", "emb", "feat", "cl1", "cl2", "G_lossf", "batch_loss"):
        ref = float(fx[k])
        assert abs(float(L[k]) - ref) <= 1e-4 * abs(ref), (k, float(L[k]), ref)     # north_star: 1e-4 relative
    L["batch_loss"].backward()
    named = dict(tr.model.named_parameters())
    for k in fx.files:
        if k.startswith("g.") and k != "g.weight_dict.w_q":
            H.check_grad(named[k[2:]].grad, fx[k], 2e-4, k)


def test_trainer_batches_and_empty_graph_steady_state(tmp_path):
    """Three batches of the real loop: graphs rebuilt at idx=1 (k = int(96*1e-4) = 0 -> empty), so
    from then on InfoNCE == -log(1/(2B-1)+1e-8); then a validation pass."""
    tr = _trainer(tmp_path, drop_rate=0.2)
    dg = tr.data_generator
    cls = []
    for idx in range(3):
        tr.model.train()
        users, pos, neg = dg.sample()
        tr._discriminator_step(users)
        bl, mf, emb, reg, cl, gl = tr._generator_step(idx, users, pos, neg)
        assert math.isfinite(float(bl))
        cls.append(float(cl))
    assert tr.image_ui_graph._nnz() == 0 and tr.text_iu_graph._nnz() == 0
    B = 48
    assert abs(cls[2] - 2 * (-math.log(1.0 / (2 * B - 1) + 1e-8))) < 1e-4
    ret = tr.test(list(dg.val_set.keys()), is_val=True)
    assert ret["recall"].shape == (3,) and 0.0 <= ret["recall"][1] <= 1.0


def test_train_step_with_injected_dropout_matches_oracle(tmp_path):
    """Identical parameters + identical dropout masks -> same loss and gradients as the oracle
    (dropout RNG streams cannot match across devices: SURVEY 7.2-6)."""
    fx = H.load("g8_gstep_full.npz")
    d, raw, U, I = H.dataset()
    tr = _trainer(tmp_path, drop_rate=0.2)
    tr.model.load_state_dict({k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("p.")}, strict=False)
    tr.model.train()
    gen = torch.Generator().manual_seed(1)
    km = [(torch.rand(I, 64, generator=gen) >= 0.2) for _ in range(2)]
    graphs = tuple(O.to_torch_sparse(O.csr_norm(x, True)) for x in (raw, raw.T, raw, raw.T, raw, raw.T))
    P = H.params(fx, requires_grad=True)
    cfg = O.Cfg(drop_rate=0.2, batch_size=48, layers=1, n_ui_layers=2)
    o = O.forward(P, torch.from_numpy(d["image_feat"]), torch.from_numpy(d["text_feat"]), graphs, cfg,
                  training=True, keep_masks=[k.float() for k in km])
    users, pos, neg = (torch.from_numpy(fx[k]) for k in ("users", "pos", "neg"))
    mf, emb, _ = O.bpr(o[0][users], o[1][pos], o[1][neg], 1e-5, 48)
    ref = mf + emb + O.feat_reg(o[2], o[3], o[4], o[5], I, 1e-5) + 0.03 * (
        O.infonce(o[8][users], o[6][users], 0.5) + O.infonce(o[9][users], o[6][users], 0.5))
    ref.backward()
    from mmssl_amd import ops
    og = tr.model(*tr._graphs(), keep_masks=[k.to(torch.uint8).to(DEV) for k in km])
    ug = users.to(DEV)
    mfg, embg = ops.bpr_gather(og[0], og[1], ug, pos.to(DEV), neg.to(DEV), 1e-5, 48)
    got = mfg + embg + tr.feat_reg_loss_calculation(og[2], og[3], og[4], og[5]) + 0.03 * (
        tr.batched_contrastive_loss(og[8][ug], og[6][ug]) + tr.batched_contrastive_loss(og[9][ug], og[6][ug]))
    assert abs(float(got) - float(ref)) <= 1e-4 * abs(float(ref))
    got.backward()
    named = dict(tr.model.named_parameters())
    for k in ("image_trans.weight", "image_trans.bias", "text_trans.weight", "user_id_embedding.weight",
              "item_id_embedding.weight", "weight_dict.w_self_attention_cat"):
        H.check_grad(named[k].grad, P[k].grad, 2e-4, k)


def test_hotpath_graph_replay_with_stream_overlap_matches_eager_sequential():
    """The whole step captured into a hipGraph with the forked streams of the hot node must give the same training
    trajectory as eager, single-stream execution of the same ops. drop_rate 0.2 with masks drawn inside the projection:
    every variant starts from the same generator state, so all of them see the same masks."""
    import scipy.sparse as sp
    from mmssl_amd import synth
    from mmssl_amd.graph import GraphPlan
    from mmssl_amd.hotpath import HotPathStep
    from mmssl_amd.Models import MMSSL
    U, I, E, dv, dt, B = 3000, 1700, 30000, 256, 128, 512
    _configure(drop_rate=0.2, batch_size=B, weight_size="[64, 64, 64]")
    from mmssl_amd import ops
    raw = synth.interaction_matrix(U, I, E, seed=5)
    ui, iu = synth.normalised_pair(raw)
    g = torch.Generator().manual_seed(3)
    img, txt = torch.randn(I, dv, generator=g).numpy(), torch.randn(I, dt, generator=g).numpy()
    batches = [(torch.randperm(U, generator=g)[:B], torch.randint(0, I, (B,), generator=g),
                torch.randint(0, I, (B,), generator=g)) for _ in range(4)]

    def run(overlap, capture, eager_loss=True):
        """One warm-up step on batch 0 (eager; inside capture() for the captured run), then the 4 batches."""
        torch.manual_seed(11)
        ops.seed_dropout(11)
        model = MMSSL(U, I, 64, [64] * 3, [0.1] * 3, img, txt).to(DEV).train()
        e1, e2 = GraphPlan(sp.csr_matrix((U, I), dtype=np.float32)), GraphPlan(sp.csr_matrix((I, U), dtype=np.float32))
        step = HotPathStep(model, (GraphPlan(ui), GraphPlan(iu), e1, e2, e1, e2), B, overlap=overlap, eager_loss=eager_loss)
        step.set_batch(*[t.to(DEV) for t in batches[0]])
        if capture:            # one executed warm-up step (the capture pass itself only records)
            assert step.capture(warmup=1), getattr(step, "capture_error", "")
        else:
            step.step()
        losses = []
        for b in batches:
            step.set_batch(*[t.to(DEV) for t in b])
            step.run()
            torch.cuda.synchronize()
            losses.append(float(step.loss))
        return losses, model.item_id_embedding.weight.detach().cpu().clone(), model.image_trans.weight.detach().cpu().clone()

    ref_l, ref_e, ref_w = run(overlap=False, capture=False)          # one stream, eager
    got_l, got_e, got_w = run(overlap=True, capture=True)            # forked streams inside a hipGraph
    ov_l, ov_e, ov_w = run(overlap=True, capture=False)              # forked streams, eager
    # the loss section as autograd launches it (separate loss kernels, backward after the loss assembly) instead of
    # the single chain with the gradients launched in the forward (ops._BatchLosses._forward_eager)
    ag_l, ag_e, ag_w = run(overlap=True, capture=True, eager_loss=False)
    assert all(np.isfinite(got_l)) and all(np.isfinite(ref_l))
    # Streams and graph replay change WHEN kernels run, never what they compute (every reduction has a fixed
    # order): the three trajectories agree to rounding. In a replayed graph the chains really overlap on the
    # device (SpMMs on one plan at the same time), so this is also the race check for shared kernel state.
    for name, (l, e, w) in (("graph+streams", (got_l, got_e, got_w)), ("streams", (ov_l, ov_e, ov_w)),
                            ("graph+streams, autograd loss section", (ag_l, ag_e, ag_w))):
        for a, b in zip(l, ref_l):
            assert abs(a - b) <= 1e-5 * abs(b), (name, l, ref_l)
        assert H.rel_err(e, ref_e) < 1e-5, name
        assert H.rel_err(w, ref_w) < 1e-5, name


def test_eval_on_device_matches_reference_recall(tmp_path):
    """G7 through the device path of test_torch (GPU scores + stable sort): Recall/NDCG/precision/hit@K
    identical to the reference, including the exact-tie items."""
    from mmssl_amd import config
    from mmssl_amd.utility import batch_test
    root = H.write_dataset_dir(str(tmp_path))
    config.configure([], data_path=root, dataset="tiny", batch_size=48)
    data = batch_test.init_data()
    g = H.load("g7_eval.npz")
    ua, ia = torch.from_numpy(g["ua"]).to(DEV), torch.from_numpy(g["ia"]).to(DEV)
    for nm, is_val in (("val", True), ("test", False)):
        users = [int(u) for u in g[nm + ".users"]]
        res = batch_test.test_torch(ua, ia, users, is_val, data=data)
        for k in ("precision", "recall", "ndcg", "hit_ratio"):
            np.testing.assert_allclose(res[k], g["%s.%s" % (nm, k)], rtol=1e-12, atol=1e-15, err_msg=k)


def test_trainer_captured_hot_path_matches_eager_trainer(tmp_path):
    """Trainer.train_batch switches to the two captured segments (hotpath.SplitHotPath) once the modal graphs are
    the cached empty plans (batch 2 on under the defaults). The whole trajectory — every batch loss and the final
    parameters — must equal the op-by-op Trainer's from the same seed: capturing (warm-up included) must not advance
    training, and the GAN gradient must reach the captured backward."""
    import os

    def run(flag, sub):
        os.environ["MMSSL_TRAINER_GRAPH"] = flag
        try:
            tr = _trainer(tmp_path / sub, drop_rate=0.2)
            dg = tr.data_generator
            losses, used = [], []
            for idx in range(7):
                tr.model.train()
                users, pos, neg = dg.sample()
                out = tr.train_batch(idx, users, pos, neg)
                losses.append([float(out[0]), float(out[1]), float(out[2]), float(out[4]), float(out[5])])
                used.append(getattr(tr, "_split", None) not in (None, False))
            torch.cuda.synchronize()
            P = {k: v.detach().cpu().clone() for k, v in tr.model.named_parameters()
                 if k.split(".")[0] in ("image_trans", "text_trans", "user_id_embedding", "item_id_embedding")}
            return np.array(losses), used, P
        finally:
            os.environ.pop("MMSSL_TRAINER_GRAPH", None)
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    ref_l, ref_used, ref_P = run("0", "a")
    got_l, got_used, got_P = run("1", "b")
    assert not any(ref_used) and got_used[-1] and got_used.count(True) >= 4, (ref_used, got_used)
    np.testing.assert_allclose(got_l, ref_l, rtol=2e-5, atol=1e-7)
    for k in ref_P:
        assert H.rel_err(got_P[k], ref_P[k]) < 2e-5, k


# ---------------------------------------------------------------------------------------------------
# f-4: the LATTICE / MICRO baselines (mmssl_amd/baselines.py) against goldens recorded from the reference classes
# ---------------------------------------------------------------------------------------------------
_BASELINE_FILES = {"lattice": "g10_lattice_lightgcn.npz", "micro": "g11_micro_lightgcn.npz",
                   "lattice_ngcf": "g13_lattice_ngcf.npz", "ngcf": "g13_ngcf.npz",
                   "micro_ngcf_sparse": "g14_micro_ngcf_sparse.npz"}


def _baseline_case(name):
    fx = H.load(_BASELINE_FILES[name])
    U, I = int(fx["n_users"]), int(fx["n_items"])
    A = sp.csr_matrix((fx["adj_val"], (fx["adj_row"], fx["adj_col"])), shape=(U + I, U + I))
    from mmssl_amd.graph import GraphPlan
    return fx, U, I, GraphPlan(A)


def _cot(U, I, D=64):
    i_ = np.arange(U + I, dtype=np.float64)[:, None]
    j_ = np.arange(D, dtype=np.float64)[None, :]
    return torch.from_numpy(np.sin(0.37 * i_ + 1.3 * j_).astype(np.float32)).to(DEV)


@pytest.mark.parametrize("name", ["lattice", "micro", "lattice_ngcf", "ngcf", "micro_ngcf_sparse"])
def test_baselines_match_reference_classes(name):
    """G10 / G11 / G13 / G14: forward(adj, build_item_graph=True) of the reference's LATTICE / MICRO (lightgcn and ngcf;
    MICRO's ngcf golden recorded on its --sparse 1 path) and its NGCF class on a tiny problem: every output, MICRO's
    contrastive loss, and the gradients - through the LEARNED item graph (projection weights, modal weights / attention
    query, id embeddings) and through the NGCF layers' transforms."""
    from mmssl_amd import baselines
    fx, U, I, plan = _baseline_case(name)
    cf = "ngcf" if "ngcf" in name else "lightgcn"
    drops = [0.0, 0.0] if cf == "ngcf" else [0.1, 0.1]
    if name == "ngcf":
        model = baselines.NGCF(U, I, 64, [64, 64], drops)
    else:
        cls = baselines.LATTICE if name.startswith("lattice") else baselines.MICRO
        model = cls(U, I, 64, [64, 64], drops, fx["image_feat"], fx["text_feat"], topk=int(fx["topk"]), cf_model=cf)
    model.load_state_dict({k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("p.")}, strict=True)
    model = model.to(DEV).train()
    outs = model(plan, build_item_graph=True)
    names = ["ua", "ia"] if not name.startswith("micro") else ["ua", "ia", "image_item", "text_item", "h"]
    assert len(outs) == len(names)
    for n, o in zip(names, outs):
        assert tuple(o.shape) == tuple(fx["o." + n].shape), (n, o.shape)
        assert H.rel_err(o.detach().cpu(), fx["o." + n]) < 2e-5, (n, H.rel_err(o.detach().cpu(), fx["o." + n]))
    cot = _cot(U, I, outs[0].shape[1])
    scalar = (outs[0] * cot[:U]).sum() + (outs[1] * cot[U:]).sum()
    if name.startswith("micro"):
        cl = model.batched_contrastive_loss(outs[2], outs[4]) + model.batched_contrastive_loss(outs[3], outs[4])
        assert abs(float(cl) - float(fx["cl"])) <= 1e-4 * abs(float(fx["cl"]))
        scalar = scalar + 0.03 * cl
    assert abs(float(scalar) - float(fx["scalar"])) <= 1e-4 * abs(float(fx["scalar"]))
    scalar.backward()
    checked = 0
    for k in fx.files:
        if k.startswith("g."):
            H.check_grad(model.get_parameter(k[2:]).grad, fx[k], 1e-4, k)               # observed: up to 2.5e-6
            checked += 1
    assert checked >= (10 if cf == "ngcf" else 6)
    if name == "ngcf":
        return
    # a second forward without rebuilding keeps the (detached) graph: same values, no gradient into the projections
    model.zero_grad()
    outs2 = model(plan, build_item_graph=False)
    assert H.rel_err(outs2[1].detach().cpu(), fx["o.ia"]) < 2e-5
    outs2[1].sum().backward()
    assert model.image_trs.weight.grad is None or float(model.image_trs.weight.grad.abs().max()) == 0.0


def test_ell_spmm_and_ngcf_tail_match_torch():
    """ops.ell_spmm (kNN-list product: forward, SDDMM gradient of the weights, scattered gradient of h), ops.mul and
    ops.ngcf_combine (leaky_relu + sum + dropout + normalize) against the same expressions in torch autograd on the CPU."""
    from mmssl_amd import ops
    gen = torch.Generator().manual_seed(5)
    N, k, d = 777, 10, 64
    idx = torch.randint(0, N, (N, k), generator=gen)
    w = torch.rand(N, k, generator=gen).requires_grad_(True)
    h = torch.randn(N, d, generator=gen).requires_grad_(True)
    cot = torch.randn(N, d, generator=gen)
    ref = (w.unsqueeze(-1) * h[idx]).sum(1)
    ref.backward(cot)
    wd, hd = w.detach().to(DEV).requires_grad_(True), h.detach().to(DEV).requires_grad_(True)
    y = ops.ell_spmm(idx.to(DEV), wd, hd)
    y.backward(cot.to(DEV))
    assert H.rel_err(y.detach().cpu(), ref.detach()) < 2e-6
    assert H.rel_err(wd.grad.cpu(), w.grad) < 2e-6 and H.rel_err(hd.grad.cpu(), h.grad) < 5e-6
    # NGCF tail with an injected dropout mask
    a, b = torch.randn(N, d, generator=gen).requires_grad_(True), torch.randn(N, d, generator=gen).requires_grad_(True)
    keep = (torch.rand(N, d, generator=gen) >= 0.3)
    c1, c2 = torch.randn(N, d, generator=gen), torch.randn(N, d, generator=gen)
    F_ = torch.nn.functional
    m = a * b
    ego = (F_.leaky_relu(a) + F_.leaky_relu(m)) * keep.float() / 0.7
    norm = F_.normalize(ego, p=2, dim=1)
    ((ego * c1).sum() + (norm * c2).sum()).backward()
    ad, bd = a.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    md = ops.mul(ad, bd)
    ego_d, norm_d = ops.ngcf_combine(ad, md, keep.to(torch.uint8).to(DEV), 1.0 / 0.7)
    ((ego_d * c1.to(DEV)).sum() + (norm_d * c2.to(DEV)).sum()).backward()
    assert H.rel_err(ego_d.detach().cpu(), ego.detach()) < 2e-6 and H.rel_err(norm_d.detach().cpu(), norm.detach()) < 2e-6
    assert H.rel_err(ad.grad.cpu(), a.grad) < 5e-6 and H.rel_err(bd.grad.cpu(), b.grad) < 5e-6


def test_two_hotpath_steps_interleaved_on_two_streams_reproduce_their_solo_trajectories():
    """Re-entrancy of the Python layer: two HotPathStep objects - different models, graphs, batches and
    injected masks - stepped alternately in one process (each on its own stream) give, step for step, the losses and the
    final parameters each gives alone (to fp32 atomics' reordering). Nothing a step hands from its forward to its loss section / backward is global."""
    from mmssl_amd import ops, synth
    from mmssl_amd.graph import GraphPlan
    from mmssl_amd.hotpath import HotPathStep
    from mmssl_amd.Models import MMSSL
    _configure(drop_rate=0.2, batch_size=128, weight_size="[64, 64]")

    def make(seed, U, I, dv, dt, sparse_modal):
        g = torch.Generator().manual_seed(seed)
        raw = synth.interaction_matrix(U, I, 6 * U, seed=seed)
        ui, iu = synth.normalised_pair(raw)
        img, txt = torch.randn(I, dv, generator=g).numpy(), torch.randn(I, dt, generator=g).numpy()
        torch.manual_seed(seed)
        model = MMSSL(U, I, 64, [64] * 2, [0.1] * 2, img, txt).to(DEV).train()
        if sparse_modal:
            m = sp.csr_matrix((np.ones(64, np.float32), (np.arange(64), np.arange(64) % I)), shape=(U, I))
        else:
            m = sp.csr_matrix((U, I), dtype=np.float32)
        a, b = GraphPlan(O.csr_norm(m, True).tocsr()), GraphPlan(O.csr_norm(m.T, True).tocsr())
        step = HotPathStep(model, (GraphPlan(ui), GraphPlan(iu), a, b, a, b), 128)
        step.keep_masks = tuple((torch.rand(I, 64, generator=g) >= 0.2).to(torch.uint8).to(DEV) for _ in range(2))
        batches = [(torch.randperm(U, generator=g)[:128].to(DEV), torch.randint(0, I, (128,), generator=g).to(DEV),
                    torch.randint(0, I, (128,), generator=g).to(DEV)) for _ in range(5)]
        return step, batches

    def solo(args):
        step, batches = make(*args)
        losses = []
        for b in batches:
            step.set_batch(*b)
            step.step()
            torch.cuda.synchronize()
            losses.append(float(step.loss))
        return losses, {k: v.detach().clone() for k, v in step.model.named_parameters()}
    A, B = (1, 700, 300, 64, 96, False), (2, 500, 420, 128, 32, True)
    la, pa = solo(A)
    lb, pb = solo(B)
    (sa, ba), (sb, bb) = make(*A), make(*B)
    ga, gb = [], []
    for k in range(5):                                   # interleaved, no synchronisation between the two objects
        sa.set_batch(*ba[k])
        sb.set_batch(*bb[k])
        sa.step()
        sb.step()
        with torch.cuda.stream(sa.stream):            # the loss buffer is written on the step's own stream
            ga.append(sa.loss.clone())
        with torch.cuda.stream(sb.stream):
            gb.append(sb.loss.clone())
    torch.cuda.synchronize()
    # Not bit-exact: the loss backward scatter-adds with fp32 atomics, whose order differs from run to run; where a
    # gradient entry nearly cancels, AdamW's g / sqrt(v) turns that last-bit noise into a visible difference of a few
    # entries after a handful of steps (observed: losses to 1e-7, parameters to 2.5e-6 of their largest entry).
    # Interference between the two objects would be gross.
    for got, want in ((ga, la), (gb, lb)):
        for x, y in zip(got, want):
            assert abs(float(x) - y) <= 2e-5 * abs(y), (ga, la, gb, lb)
    worst = 0.0
    for step, want in ((sa, pa), (sb, pb)):
        for k, v in step.model.named_parameters():
            e = H.rel_err(v.detach().cpu(), want[k].cpu())
            worst = max(worst, e)
            assert e < 1e-4, (k, e)               # observed: up to 2.5e-6
    print("interleaved vs solo: worst loss dev %.2e, worst parameter dev %.2e" % (
        max(abs(float(x) - y) / abs(y) for got, want in ((ga, la), (gb, lb)) for x, y in zip(got, want)), worst))


def test_hotpath_batch_ring_equals_set_batch():
    """HotPathStep.set_batch_ring: every step picks slot (completed optimiser steps mod n) of a device-resident ring by a
    launch inside the (captured) step. Same trajectory as set_batch() of the same slot before every replay."""
    import scipy.sparse as sp
    from mmssl_amd import ops, synth
    from mmssl_amd.graph import GraphPlan
    from mmssl_amd.hotpath import HotPathStep
    from mmssl_amd.Models import MMSSL
    U, I, E, dv, dt, B = 2000, 1200, 20000, 128, 64, 256
    _configure(drop_rate=0.2, batch_size=B, weight_size="[64, 64]")
    raw = synth.interaction_matrix(U, I, E, seed=6)
    ui, iu = synth.normalised_pair(raw)
    g = torch.Generator().manual_seed(4)
    img, txt = torch.randn(I, dv, generator=g).numpy(), torch.randn(I, dt, generator=g).numpy()
    ring = torch.stack([torch.stack([torch.randperm(U, generator=g)[:B], torch.randint(0, I, (B,), generator=g),
                                     torch.randint(0, I, (B,), generator=g)]) for _ in range(3)]).to(DEV)

    def run(use_ring):
        torch.manual_seed(12)
        ops.seed_dropout(12)
        model = MMSSL(U, I, 64, [64] * 2, [0.1] * 2, img, txt).to(DEV).train()
        e1, e2 = GraphPlan(sp.csr_matrix((U, I), dtype=np.float32)), GraphPlan(sp.csr_matrix((I, U), dtype=np.float32))
        step = HotPathStep(model, (GraphPlan(ui), GraphPlan(iu), e1, e2, e1, e2), B)
        if use_ring:
            step.set_batch_ring(ring)
        else:
            step.set_batch(ring[0])
        assert step.capture(warmup=1), getattr(step, "capture_error", "")      # one executed step: slot 0
        losses = []
        for k in range(1, 8):
            if not use_ring:
                step.set_batch(ring[k % 3])
            step.run()
            torch.cuda.synchronize()
            losses.append(float(step.loss))
            # the deterministic half of the claim: the step read exactly slot (completed steps mod 3) - index buffers are
            # integers, no atomics involved
            assert torch.equal(step.batch, ring[k % 3]), k
            assert int(step._steps_done) == k + 1
        return losses, model.user_id_embedding.weight.detach().cpu().clone()
    la, ea = run(True)
    lb, eb = run(False)
    assert len(set(round(x, 6) for x in la)) > 3           # the batches really change
    for a, b in zip(la, lb):
        assert abs(a - b) <= 1e-5 * abs(b), (la, lb)
    # (two runs of the SAME step sequence: what separates them is the order of the BPR / InfoNCE scatter atomics, ~1e-9 on a
    # gradient, which AdamW's first steps - update ~ g / |g| - magnify; observed 2e-6 .. 1.01e-5 over the rounds)
    assert H.rel_err(ea, eb) < 3e-5


# ---------------------------------------------------------------------------------------------------
# G12: the reference's own Trainer.train() for 8 batches + its evaluation, recorded by oracle/gen_golden.py g12
# (every random tensor of the loop injected): K-step trajectory and Recall@20 parity of the product Trainer
# ---------------------------------------------------------------------------------------------------
def _g12_run(tmp_path, graph_flag):
    import os
    from mmssl_amd import config
    from mmssl_amd.utility import batch_test
    fx = H.load("g12_train_trajectory.npz")
    n = int(fx["n_batches"])
    root = H.write_dataset_dir(str(tmp_path))
    _configure(data_path=root, dataset="tiny", drop_rate=0.0, G_drop1=0.0, G_drop2=0.0, m_topk_rate=float(fx["m_topk_rate"]),
               T=1, epoch=1)
    dg = batch_test.init_data()
    from mmssl_amd.main import Trainer, set_seed
    set_seed(2022)
    os.environ["MMSSL_TRAINER_GRAPH"] = graph_flag
    try:
        tr = Trainer(data_config={})
        # the sampler after set_seed + Trainer(): the same batches the reference's loop drew
        for b in range(n):
            u, p, q = dg.sample()
            assert [int(x) for x in u] == fx["b%d.users" % b].tolist() and [int(x) for x in p] == fx["b%d.pos" % b].tolist() \
                and [int(x) for x in q] == fx["b%d.neg" % b].tolist(), b
        missing = tr.model.load_state_dict({k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("m0.")}, strict=False)
        assert all(k.startswith(("encoder.", "align.")) for k in missing.missing_keys), missing
        tr.D.load_state_dict({k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("D0.")})
        cur = {"b": 0}
        tr.noise_hook = lambda kind, shape: torch.from_numpy(fx["b%d.%s" % (cur["b"], "gumbel_u" if kind == "gumbel" else "gp_alpha")]).reshape(shape)
        rows, used = [], []
        for b in range(n):
            cur["b"] = b
            tr.model.train()
            users, pos, neg = (fx["b%d.%s" % (b, k)].tolist() for k in ("users", "pos", "neg"))
            out = tr.train_batch(b, users, pos, neg)
            rows.append([float(out[0]), float(out[1]), float(out[2]), float(out[4]), float(out[5])])
            used.append(getattr(tr, "_split", None) not in (None, False))
        torch.cuda.synchronize()
        want = np.array([[float(fx["b%d.%s" % (b, k)]) for k in ("batch_loss", "mf", "emb")]
                         + [float(fx["b%d.cl1" % b]) + float(fx["b%d.cl2" % b]), float(fx["b%d.G_lossf" % b])] for b in range(n)])
        got = np.array(rows)
        P = {k: v.detach().cpu() for k, v in tr.model.state_dict().items()}
        assert tr.image_ui_graph._nnz() == int(fx["final.img_ui_nnz"]) == 0
        tr.model.eval()
        with torch.no_grad():
            outs = tr.model(*tr._graphs())
        ev = {}
        for nm, is_val in (("val", True), ("test", False)):
            ev[nm] = tr.test([int(u) for u in fx[nm + ".users"]], is_val)
        return fx, got, want, used, P, outs[0].cpu(), outs[1].cpu(), ev
    finally:
        os.environ.pop("MMSSL_TRAINER_GRAPH", None)


@pytest.mark.parametrize("graph_flag", ["0", "1"])
def test_g12_trainer_trajectory_and_recall_match_reference(tmp_path, graph_flag):
    """8 batches of the REFERENCE loop (batches 0-1 on the interaction graph, 2 on the top-1 graph, 3+ on empty modal
    graphs; discriminator step, gradient penalty, generator step, AdamW / Adam) against the product Trainer with the
    reference's Gumbel uniforms and penalty alphas injected: every batch loss component 1e-4, the final parameters
    1e-4, the eval-mode embeddings 1e-4, and Recall / NDCG / precision / hit @ 10, 20, 50 EXACTLY - op by op
    (graph_flag 0) and on the captured hot path (SplitHotPath from the fifth batch on)."""
    fx, got, want, used, P, ua, ia, ev = _g12_run(tmp_path, graph_flag)
    if graph_flag == "1":
        assert used[-1] and used.count(True) >= 3, used
    else:
        assert not any(used)
    np.testing.assert_allclose(got[:, :4], want[:, :4], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(got[:, 4], want[:, 4], rtol=1e-4, atol=1e-5)           # -mean(100 * sigmoid(.))
    # weight_dict.w_q: an all-zero gradient in the reference, i.e. AdamW's weight decay only (8 x lr x 0.01 of its value);
    # the product's optimiser hands it the same zero gradient (Models.MMSSL.__init__) - a trained checkpoint round-trips
    for k in ("image_trans.weight", "image_trans.bias", "text_trans.weight", "text_trans.bias", "user_id_embedding.weight",
              "item_id_embedding.weight", "weight_dict.w_self_attention_cat", "weight_dict.w_q"):
        e = H.rel_err(P[k], fx["m1." + k])
        assert e < 1e-4, (k, e)
        moved = H.rel_err(fx["m0." + k], fx["m1." + k])
        assert moved > 10 * e, (k, moved, e)                     # the comparison is not vacuous: training moved it
    for k in ("weight_dict.w_k", "weight_dict.w_v"):             # no gradient at all on either side: untouched
        assert np.array_equal(P[k].numpy(), fx["m1." + k]) and np.array_equal(fx["m0." + k], fx["m1." + k]), k
    assert H.rel_err(ua, fx["eval.ua"]) < 1e-4 and H.rel_err(ia, fx["eval.ia"]) < 1e-4
    for nm in ("val", "test"):
        for k in ("precision", "recall", "ndcg", "hit_ratio"):
            np.testing.assert_allclose(ev[nm][k], fx["%s.%s" % (nm, k)], rtol=1e-12, atol=1e-15, err_msg=nm + k)


# ---------------------------------------------------------------------------------------------------
# G15: the reference's own Trainer.train() of LATTICE and MICRO (3 epochs x 7 batches on the tiny dataset), recorded
# by oracle/gen_golden_baselines.py trainer: the product loop (mmssl_amd/baselines_main.py) on the HIP models
# ---------------------------------------------------------------------------------------------------
G15_ARGV = ["--dataset", "tiny", "--batch_size", "128", "--epoch", "3", "--verbose", "1", "--topk", "10", "--seed", "7",
            "--Ks", "[10, 20]", "--lr", "0.005"]


@pytest.mark.parametrize("which", ["lattice", "micro"])
def test_g15_baseline_trainers_match_reference(tmp_path, which):
    """21 batches of the REFERENCE loop (item graph rebuilt in the first batch of every epoch and detached afterwards,
    Adam with a per-epoch LambdaLR, MICRO's two contrastive terms, validation + test after every epoch) against
    baselines_main.Trainer with the reference's initial parameters and its sampled batches replayed: every batch's loss
    terms 1e-4, the final parameters 1e-4 (training moved them 10x further than that), the eval-mode embeddings 1e-4,
    and precision / recall / NDCG / hit @ 10, 20 of all six evaluations EXACTLY."""
    from mmssl_amd import baselines_main as BM
    from mmssl_amd.utility.load_data import Data
    fx = H.load("g15_%s_trainer.npz" % which)
    extra = ["--sparse", "1"] if which == "micro" else []
    assert " ".join(G15_ARGV + extra) == str(fx["argv"])
    root = H.write_dataset_dir(str(tmp_path))
    a = BM.parse_args(which, ["--data_path", root] + G15_ARGV + extra)
    data = Data(path=root + "tiny", batch_size=a.batch_size)
    _, norm_adj, _ = data.get_adj_mat()
    BM.set_seed(a.seed)
    tr = BM.Trainer({"n_users": data.n_users, "n_items": data.n_items, "norm_adj": norm_adj}, a, data=data)
    tr.model.load_state_dict({k[3:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("m0.")}, strict=True)
    n = int(fx["n_batches"])
    # the sampler is pinned bit for bit by G6; its stream position after the constructor depends on how many draws the
    # parameter initialisation took, so the recorded batches are replayed
    cur = {"b": 0}

    def sample():
        b = cur["b"]
        cur["b"] += 1
        return [fx["b%d.%s" % (b, k)].tolist() for k in ("users", "pos", "neg")]
    data.sample = sample
    rows, inner = [], tr._batch_losses

    def batch_losses(outs, users, pos, neg):
        out = inner(outs, users, pos, neg)
        rows.append([float(out[0].detach()), float(out[1].detach())] + ([float(out[3].detach())] if which == "micro" else []))
        return out
    tr._batch_losses = batch_losses
    evals, inner_test = [], tr.test

    def test(users, is_val):
        ret = inner_test(users, is_val)
        evals.append((bool(is_val), [int(u) for u in users], ret))
        return ret
    tr.test = test
    tr.train()
    torch.cuda.synchronize()
    assert cur["b"] == n == len(rows)
    keys = ("mf", "emb") + (("cl",) if which == "micro" else ())
    want = np.array([[float(fx["b%d.%s" % (b, k)]) for k in keys] for b in range(n)])
    np.testing.assert_allclose(np.array(rows), want, rtol=1e-4, atol=1e-8)
    P = {k: v.detach().cpu() for k, v in tr.model.state_dict().items()}
    moved_any = 0
    for k in P:
        e = H.rel_err(P[k], fx["m1." + k])
        assert e < 1e-4, (k, e)
        moved = H.rel_err(fx["m0." + k], fx["m1." + k])
        if moved > 10 * 1e-4:
            moved_any += 1
            assert moved > 10 * e, (k, moved, e)
    assert moved_any >= 6, moved_any
    assert abs(tr.optimizer.param_groups[0]["lr"] - float(fx["final_lr"])) <= 1e-12
    assert len(evals) == int(fx["n_evals"]) == 6
    for e, (is_val, users, ret) in enumerate(evals):
        assert is_val == bool(fx["e%d.is_val" % e]) and users == fx["e%d.users" % e].tolist()
        for k in ("precision", "recall", "ndcg", "hit_ratio"):
            np.testing.assert_allclose(ret[k], fx["e%d.%s" % (e, k)], rtol=1e-12, atol=1e-15, err_msg="eval %d %s" % (e, k))
    tr.model.eval()
    with torch.no_grad():
        outs = tr.model(tr.norm_adj, build_item_graph=True)
    assert H.rel_err(outs[0].cpu(), fx["eval.ua"]) < 1e-4 and H.rel_err(outs[1].cpu(), fx["eval.ia"]) < 1e-4
