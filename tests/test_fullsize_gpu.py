"""Parity at BASELINE.json's full sizes: the Tiktok shape directly against the oracle (seconds on CPU),
the Amazon-Baby shape through size-independent properties (adjointness of the backward SpMM,
linearity, softmax rows, determinism, loss invariants)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import helpers as H
import mmssl_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(shape, G=3):
    from mmssl_amd import config, synth
    from mmssl_amd.graph import GraphPlan
    U, I, E, dv, dt = synth.SHAPES[shape]
    config.configure([], drop_rate=0.0, batch_size=1024, weight_size=str([64] * G), debug=True)
    raw = synth.interaction_matrix(U, I, E, seed=1)
    ui, iu = synth.normalised_pair(raw)
    return U, I, dv, dt, raw, ui, iu, GraphPlan(ui), GraphPlan(iu)


def test_tiktok_full_size_forward_and_losses_match_oracle():
    """configs[1]: Tiktok (9319 x 6710, 59.5K edges, V128/T768), d=64, 3-layer GCN + V/T InfoNCE."""
    from mmssl_amd import ops
    from mmssl_amd.Models import MMSSL
    U, I, dv, dt, raw, ui, iu, P_ui, P_iu = _setup("tiktok")
    g = torch.Generator().manual_seed(0)
    img, txt = torch.randn(I, dv, generator=g), torch.randn(I, dt, generator=g)
    torch.manual_seed(4)
    model = MMSSL(U, I, 64, [64] * 3, [0.1] * 3, img.numpy(), txt.numpy())
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    model = model.to(DEV).train()
    rng = np.random.default_rng(1)
    # one sparse modal graph pair so that the InfoNCE first operand is not all-zero
    us = rng.choice(U, 1024, replace=False)
    modal = sp.csr_matrix((np.ones(1024, np.float32), (us, rng.integers(0, I, 1024))), shape=(U, I))
    from mmssl_amd.graph import GraphPlan
    m_ui, m_iu = O.csr_norm(modal, True).tocsr(), O.csr_norm(modal.T, True).tocsr()
    graphs_g = (P_ui, P_iu, GraphPlan(m_ui), GraphPlan(m_iu), GraphPlan(m_ui), GraphPlan(m_iu))
    A = [O.to_torch_sparse(x) for x in (ui, iu, m_ui, m_iu, m_ui, m_iu)]
    users = torch.from_numpy(us)
    pos = torch.from_numpy(rng.integers(0, I, 1024))
    neg = torch.from_numpy(rng.integers(0, I, 1024))
    cfg = O.Cfg(drop_rate=0.0, n_ui_layers=3, batch_size=1024)
    o = O.forward(P, img, txt, A, cfg, training=False)
    mf, emb, _ = O.bpr(o[0][users], o[1][pos], o[1][neg], 1e-5, 1024)
    ref = mf + emb + O.feat_reg(o[2], o[3], o[4], o[5], I, 1e-5) + 0.03 * (
        O.infonce(o[8][users], o[6][users], 0.5) + O.infonce(o[9][users], o[6][users], 0.5))
    ref.backward()
    og = model(*graphs_g)
    for k in (0, 1, 2, 4, 8, 10):
        assert H.rel_err(og[k].detach().cpu(), o[k].detach()) < 1e-4, k
    t = ops.batch_losses_vec(og[0], og[1], og[8], og[9], users.to(DEV), pos.to(DEV), neg.to(DEV), 1e-5, 1024, 0.5)
    w = torch.tensor([1.0, 1.0, 1.0, 0.03, 0.03], device=DEV)
    got = ops.loss_assemble(t, w, model.feat_sumsq(og[2], og[3], og[4], og[5]), 1e-5 * 0.5 / I)
    assert abs(float(got) - float(ref)) <= 1e-4 * abs(float(ref)), (float(got), float(ref))     # north_star bar
    got.backward()
    named = dict(model.named_parameters())
    for k in ("image_trans.weight", "text_trans.weight", "user_id_embedding.weight", "item_id_embedding.weight",
              "weight_dict.w_self_attention_cat"):
        H.check_grad(named[k].grad, P[k].grad, 5e-4, k)


def test_baby_full_size_properties():
    """configs[2]: Amazon-Baby shape (35598 x 18357, 256308 edges), d=64."""
    from mmssl_amd import ops
    U, I, dv, dt, raw, ui, iu, P_ui, P_iu = _setup("baby")
    g = torch.Generator().manual_seed(0)
    X = torch.randn(I, 64, generator=g).to(DEV)
    X2 = torch.randn(I, 64, generator=g).to(DEV)
    Yv = torch.randn(U, 64, generator=g).to(DEV)
    Y = ops.spmm(P_ui, X)
    # adjointness: <A x, y> == <x, A^T y>  (the backward SpMM really is the transpose)
    lhs = float((Y.double() * Yv.double()).sum())
    rhs = float((X.double() * ops.spmm(P_ui, Yv, transpose=True).double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * abs(lhs)
    # linearity
    Z = ops.spmm(P_ui, 2.0 * X - 3.0 * X2)
    assert H.rel_err(Z.cpu(), (2.0 * Y - 3.0 * ops.spmm(P_ui, X2)).cpu()) < 1e-5
    # row sums of A_ui with 1/sqrt(deg) values: A.1 = sqrt(deg) (empty rows 0)
    ones = torch.ones(I, 64, device=DEV)
    deg = np.diff(raw.indptr).astype(np.float64)
    np.testing.assert_allclose(ops.spmm(P_ui, ones)[:, 0].cpu().numpy(), np.sqrt(deg), rtol=2e-6, atol=1e-6)
    # fused softmax rows sum to one; bitwise determinism
    S = ops.spmm(P_iu, Yv, epilogue=ops.EPI_SOFTMAX)
    assert float((S.sum(1) - 1).abs().max()) < 1e-5 and float(S.min()) >= 0.0
    assert torch.equal(S, ops.spmm(P_iu, Yv, epilogue=ops.EPI_SOFTMAX))
    # against the oracle on a 2000-row sample of the full product
    ref = O.spmm(O.to_torch_sparse(ui), X.cpu())
    rows = torch.randperm(U, generator=g)[:2000]
    assert H.rel_err(Y.cpu()[rows], ref[rows]) < 3e-6
    # InfoNCE invariants at B=1024: symmetric inputs scale-invariant (cosine), zero z1 -> closed form
    z = torch.randn(1024, 64, generator=g).to(DEV)
    z2 = torch.randn(1024, 64, generator=g).to(DEV)
    a = float(ops.infonce(z, z2, 0.5))
    b = float(ops.infonce(7.5 * z, 0.01 * z2, 0.5))
    assert abs(a - b) <= 2e-6 * abs(a)
    zero = float(ops.infonce(torch.zeros_like(z), z2, 0.5))
    assert abs(zero + np.log(1.0 / (2 * 1024 - 1) + 1e-8)) < 1e-5


# ---------------------------------------------------------------------------------------------------
# configs[2] end to end: the FULL Amazon-Baby hot-path step (18357 x 4096 projection, fused propagate node,
# batch losses, backward; eager AND as the captured hipGraph bench.py times) against the oracle.
# ---------------------------------------------------------------------------------------------------
_BABY = {}


def _baby_case(modal):
    """(model factory, graphs, batch, masks, oracle loss, oracle grads) for the Baby shape; `modal` = 'empty'
    (the reference's steady state, what bench.py times) or 'full' (its first two batches: modal graphs ARE the
    interaction graph — the same GraphPlan objects passed twice, like Trainer's initial state)."""
    if modal in _BABY:
        return _BABY[modal]
    from mmssl_amd.graph import GraphPlan
    from mmssl_amd.Models import MMSSL
    U, I, dv, dt, raw, ui, iu, P_ui, P_iu = _setup("baby")
    from mmssl_amd import config
    config.configure([], drop_rate=0.2, batch_size=1024, weight_size=str([64] * 3), debug=True)
    g = torch.Generator().manual_seed(0)
    img, txt = torch.randn(I, dv, generator=g), torch.randn(I, dt, generator=g)
    torch.manual_seed(4)
    cpu_model = MMSSL(U, I, 64, [64] * 3, [0.1] * 3, img.numpy(), txt.numpy())
    state0 = {k: v.detach().clone() for k, v in cpu_model.state_dict().items()}
    del cpu_model
    km = [(torch.rand(I, 64, generator=g) >= 0.2) for _ in range(2)]
    rng = np.random.default_rng(1)
    users = torch.from_numpy(rng.choice(U, 1024, replace=False))
    pos = torch.from_numpy(rng.integers(0, I, 1024))
    neg = torch.from_numpy(rng.integers(0, I, 1024))
    if modal == "empty":
        e_ui, e_iu = sp.csr_matrix((U, I), dtype=np.float32), sp.csr_matrix((I, U), dtype=np.float32)
        graphs_g = (P_ui, P_iu, GraphPlan(e_ui), GraphPlan(e_iu), GraphPlan(e_ui), GraphPlan(e_iu))
        mats = (ui, iu, e_ui, e_iu, e_ui, e_iu)
    else:
        graphs_g = (P_ui, P_iu, P_ui, P_iu, P_ui, P_iu)
        mats = (ui, iu, ui, iu, ui, iu)
    A = [O.to_torch_sparse(x) for x in mats]
    P = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in state0.items()
         if not k.startswith(("image_embedding", "text_embedding", "batch_norm", "encoder.", "align."))}
    cfg = O.Cfg(drop_rate=0.2, n_ui_layers=3, batch_size=1024)
    o = O.forward(P, img, txt, A, cfg, training=True, keep_masks=[k.float() for k in km])
    mf, emb, _ = O.bpr(o[0][users], o[1][pos], o[1][neg], 1e-5, 1024)
    ref = mf + emb + O.feat_reg(o[2], o[3], o[4], o[5], I, 1e-5) + 0.03 * (
        O.infonce(o[8][users], o[6][users], 0.5) + O.infonce(o[9][users], o[6][users], 0.5))
    ref.backward()
    grads = {k: P[k].grad.clone() for k in ("image_trans.weight", "image_trans.bias", "text_trans.weight",
                                            "text_trans.bias", "user_id_embedding.weight", "item_id_embedding.weight")}
    if modal == "full":
        grads["weight_dict.w_self_attention_cat"] = P["weight_dict.w_self_attention_cat"].grad.clone()

    def make_step():
        from mmssl_amd.hotpath import HotPathStep
        model = MMSSL(U, I, 64, [64] * 3, [0.1] * 3, img.numpy(), txt.numpy())
        model.load_state_dict(state0)
        model = model.to(DEV).train()
        step = HotPathStep(model, graphs_g, 1024, decay=1e-5)
        step.keep_masks = tuple(k.to(torch.uint8).to(DEV) for k in km)
        step.set_batch(users.to(DEV), pos.to(DEV), neg.to(DEV))
        return model, step
    _BABY[modal] = (make_step, state0, float(ref), grads)
    return _BABY[modal]


def _check_baby(model, step, ref, grads, tag):
    got = float(step.loss)
    assert abs(got - ref) <= 1e-4 * abs(ref), (tag, got, ref)                       # north_star bar
    named = dict(model.named_parameters())
    for k, gref in grads.items():
        H.check_grad(named[k].grad, gref, 5e-4, (tag, k))


@pytest.mark.parametrize("modal", ["empty", "full"])
def test_baby_full_step_eager_matches_oracle(modal):
    make_step, state0, ref, grads = _baby_case(modal)
    model, step = make_step()
    step.step()
    torch.cuda.synchronize()
    _check_baby(model, step, ref, grads, "eager/" + modal)


def test_baby_full_step_captured_graph_matches_oracle():
    """The unit bench.py times: one whole step (3 forked streams, deferred wgrad join, fused AdamW) replayed from
    a hipGraph. Parameters are restored after the warm-up / capture steps, so the replay starts from the same
    state as the oracle's step; its loss and every parameter gradient must match."""
    make_step, state0, ref, grads = _baby_case("empty")
    model, step = make_step()
    assert step.capture(warmup=2), getattr(step, "capture_error", "")
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(state0[k])
    torch.cuda.synchronize()
    step.run()
    torch.cuda.synchronize()
    _check_baby(model, step, ref, grads, "captured")
    # a second replay from the same parameters: the forward (fixed-order reductions everywhere) reproduces the loss
    # bit for bit; gradients agree to rounding (the BPR backward scatter-adds duplicate batch items with hardware
    # fp32 atomics, whose order is not fixed)
    first = float(step.loss)
    g1 = model.image_trans.weight.grad.clone()
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(state0[k])
    step.run()
    torch.cuda.synchronize()
    assert float(step.loss) == first
    assert H.rel_err(model.image_trans.weight.grad.cpu(), g1.cpu()) < 1e-5


def test_tiktok_three_modalities_v_a_t_match_oracle_extension():
    """configs[1] as BASELINE.json words it: Tiktok, 3-layer GCN + V/A/T InfoNCE. The reference itself only has V and
    T (main.py:54-55); the acoustic branch follows the same pattern (oracle.forward_multi, pinned to the reference
    for V/T, UNPINNED for A). HIP ops vs that oracle: loss 1e-4, gradients 5e-4, all three projections included."""
    from mmssl_amd import config, ops
    from mmssl_amd.Models import MMSSL
    U, I, dv, dt, raw, ui, iu, P_ui, P_iu = _setup("tiktok")
    config.configure([], drop_rate=0.2, batch_size=1024, weight_size=str([64] * 3), debug=True)
    da = 128
    g = torch.Generator().manual_seed(0)
    img, txt, aud = torch.randn(I, dv, generator=g), torch.randn(I, dt, generator=g), torch.randn(I, da, generator=g)
    torch.manual_seed(4)
    model = MMSSL(U, I, 64, [64] * 3, [0.1] * 3, img.numpy(), txt.numpy(), extra_feats={"audio": aud.numpy()})
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()
         if not k.startswith(("image_embedding", "text_embedding", "audio_embedding", "batch_norm", "encoder.", "align."))}
    model = model.to(DEV).train()
    rng = np.random.default_rng(1)
    from mmssl_amd.graph import GraphPlan
    modal_m, modal_g = [], []
    for s in range(3):                      # a different sparse modal graph pair per modality
        us = rng.choice(U, 1024, replace=False)
        m = sp.csr_matrix((np.ones(1024, np.float32), (us, rng.integers(0, I, 1024))), shape=(U, I))
        m_ui, m_iu = O.csr_norm(m, True).tocsr(), O.csr_norm(m.T, True).tocsr()
        modal_m.append((O.to_torch_sparse(m_ui), O.to_torch_sparse(m_iu)))
        modal_g.append((GraphPlan(m_ui), GraphPlan(m_iu)))
    km = [(torch.rand(I, 64, generator=g) >= 0.2) for _ in range(3)]
    users = torch.from_numpy(rng.choice(U, 1024, replace=False))
    pos = torch.from_numpy(rng.integers(0, I, 1024))
    neg = torch.from_numpy(rng.integers(0, I, 1024))
    cfg = O.Cfg(drop_rate=0.2, n_ui_layers=3, batch_size=1024)
    o = O.forward_multi(P, [img, txt, aud], (O.to_torch_sparse(ui), O.to_torch_sparse(iu)), modal_m, cfg,
                        names=("image", "text", "audio"), training=True, keep_masks=[k.float() for k in km])
    ref = O.generator_loss_multi(o, users, pos, neg, I, cfg)
    ref.backward()
    out = model(P_ui, P_iu, modal_g[0][0], modal_g[0][1], modal_g[1][0], modal_g[1][1],
                keep_masks=[k.to(torch.uint8).to(DEV) for k in km], extra_graphs={"audio": modal_g[2]})
    assert len(out) == 16
    assert H.rel_err(out[0].detach().cpu(), o["ua"].detach()) < 1e-4 and H.rel_err(out[13].detach().cpu(), o["user_feats"][2].detach()) < 1e-4
    ug = users.to(DEV)
    mf, emb = ops.bpr_gather(out[0], out[1], ug, pos.to(DEV), neg.to(DEV), 1e-5, 1024)
    feat = sum(ops.sumsq(out[k]) for k in (2, 3, 4, 5, 12, 13)) * (0.5 * 1e-5 / I)
    cl = sum(ops.infonce(out[k], out[0], 0.5, idx=ug) for k in (8, 9, 14))
    got = mf + emb + feat + 0.03 * cl
    assert abs(float(got) - float(ref)) <= 1e-4 * abs(float(ref)), (float(got), float(ref))
    got.backward()
    for k in ("image_trans.weight", "text_trans.weight", "audio_trans.weight", "audio_trans.bias", "user_id_embedding.weight",
              "item_id_embedding.weight", "weight_dict.w_self_attention_cat"):
        H.check_grad(model.get_parameter(k).grad, P[k].grad, 5e-4, k)


@pytest.mark.parametrize("modal_kind", ["sparse", "empty"])
def test_tiktok_20_step_trajectory_matches_oracle_with_torch_adamw(modal_kind):
    """configs[1] as a TRAJECTORY: 20 hot-path steps (forward, BPR + 2x InfoNCE + regulariser, backward, fused AdamW) on the
    Tiktok shape against 20 steps of the oracle driven by torch.optim.AdamW on the CPU — a different batch every step,
    fixed injected dropout masks, sparse modal graphs (so the InfoNCE views and w_self_attention_cat carry gradient).
    modal_kind "empty" is the reference's steady state (what bench.py times): there the step updates the embedding tables on
    the GCN chain's side stream while the weight gradient still runs, and the projection weights inside the weight-gradient
    epilogue (HotPathStep fuse_adam) - the same trajectory as torch.optim.AdamW after the whole backward.
    Eager steps and hipGraph replays: every step's loss within 1e-4, the final parameters within 5e-4 of the largest
    entry (1/100 of their movement) and every row of the embedding tables within 1 % of its own movement."""
    from mmssl_amd.graph import GraphPlan
    from mmssl_amd.hotpath import HotPathStep
    from mmssl_amd.Models import MMSSL
    from mmssl_amd import config
    U, I, dv, dt, raw, ui, iu, P_ui, P_iu = _setup("tiktok")
    config.configure([], drop_rate=0.2, batch_size=1024, weight_size=str([64] * 3), debug=True)
    g = torch.Generator().manual_seed(0)
    img, txt = torch.randn(I, dv, generator=g), torch.randn(I, dt, generator=g)
    torch.manual_seed(4)
    cpu_model = MMSSL(U, I, 64, [64] * 3, [0.1] * 3, img.numpy(), txt.numpy())
    state0 = {k: v.detach().clone() for k, v in cpu_model.state_dict().items()}
    del cpu_model
    km = [(torch.rand(I, 64, generator=g) >= 0.2) for _ in range(2)]
    rng = np.random.default_rng(1)
    us = rng.choice(U, 1024, replace=False)
    modal = sp.csr_matrix((np.ones(1024, np.float32), (us, rng.integers(0, I, 1024))), shape=(U, I))
    if modal_kind == "empty":
        modal = sp.csr_matrix((U, I), dtype=np.float32)
    m_ui, m_iu = O.csr_norm(modal, True).tocsr(), O.csr_norm(modal.T, True).tocsr()
    A = [O.to_torch_sparse(x) for x in (ui, iu, m_ui, m_iu, m_ui, m_iu)]
    steps = 20
    batches = [(torch.from_numpy(rng.choice(U, 1024, replace=False)), torch.from_numpy(rng.integers(0, I, 1024)),
                torch.from_numpy(rng.integers(0, I, 1024))) for _ in range(steps)]
    names = ("image_trans.weight", "image_trans.bias", "text_trans.weight", "text_trans.bias", "user_id_embedding.weight",
             "item_id_embedding.weight", "weight_dict.w_self_attention_cat")
    # ---- oracle trajectory -------------------------------------------------------------------------------------
    P = {k: v.clone().requires_grad_(k in names) for k, v in state0.items()
         if not k.startswith(("image_embedding", "text_embedding", "batch_norm", "encoder.", "align."))}
    opt = torch.optim.AdamW([P[k] for k in names], lr=5.5e-4)          # main.py:76-80 (default betas / eps / weight decay)
    cfg = O.Cfg(drop_rate=0.2, n_ui_layers=3, batch_size=1024)
    ref_losses = []
    for users, pos, neg in batches:
        opt.zero_grad()
        o = O.forward(P, img, txt, A, cfg, training=True, keep_masks=[k.float() for k in km])
        mf, emb, _ = O.bpr(o[0][users], o[1][pos], o[1][neg], 1e-5, 1024)
        loss = mf + emb + O.feat_reg(o[2], o[3], o[4], o[5], I, 1e-5) + 0.03 * (
            O.infonce(o[8][users], o[6][users], 0.5) + O.infonce(o[9][users], o[6][users], 0.5))
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
    assert ref_losses[-1] != ref_losses[0]
    # ---- HIP: eager steps, then replays of the captured step from the same start -----------------------------------
    graphs_g = (P_ui, P_iu, GraphPlan(m_ui), GraphPlan(m_iu), GraphPlan(m_ui), GraphPlan(m_iu))
    for mode in ("eager", "graph"):
        model = MMSSL(U, I, 64, [64] * 3, [0.1] * 3, img.numpy(), txt.numpy())
        model.load_state_dict(state0)
        model = model.to(DEV).train()
        step = HotPathStep(model, graphs_g, 1024, decay=1e-5)
        assert step._tables_early == (modal_kind == "empty")
        step.keep_masks = tuple(k.to(torch.uint8).to(DEV) for k in km)
        if mode == "graph":
            step.set_batch(*(x.to(DEV) for x in batches[0]))
            assert step.capture(warmup=2), getattr(step, "capture_error", "")
            with torch.no_grad():
                for k, p in model.named_parameters():
                    p.copy_(state0[k])
            step.optimizer.reset_state()
        got = []
        for users, pos, neg in batches:
            step.set_batch(users.to(DEV), pos.to(DEV), neg.to(DEV))
            step.run()
            torch.cuda.synchronize()
            got.append(float(step.loss))
        np.testing.assert_allclose(got, ref_losses, rtol=1e-4, atol=0, err_msg=mode)
        named = dict(model.named_parameters())
        for k in names:
            if modal_kind == "empty" and k == "weight_dict.w_self_attention_cat":
                continue            # zero gradient: only the weight decay moves it (checked below)
            # AdamW normalises every element's step to ~lr whatever its gradient's size, so an element whose gradient
            # is small next to the fp32 rounding of either implementation may move differently by O(lr): the bound is
            # 5e-4 of the largest entry and at most 1/100 of how far training moved the tensor
            e = H.rel_err(named[k].detach().cpu(), P[k].detach())
            moved = H.rel_err(state0[k], P[k].detach())
            assert e < 5e-4 and moved > 100 * e, (mode, k, e, moved)
        if modal_kind == "empty":
            k = "weight_dict.w_self_attention_cat"
            assert H.rel_err(named[k].detach().cpu(), P[k].detach()) < 1e-6
        for k in ("user_id_embedding.weight", "item_id_embedding.weight"):
            a, b = named[k].detach().cpu().double(), P[k].detach().double()
            d0 = (b - state0[k].double()).abs().amax(1)                     # how far the oracle moved each row
            err = (a - b).abs().amax(1)
            assert float((err / (d0 + 1e-2 * float(d0.max()))).max()) < 1e-2, (mode, k)


@pytest.mark.parametrize("kind", ["d128", "narrow_text", "three_modalities"])
def test_hotpath_step_trains_every_parameter_of_models_off_the_packed_node(kind):
    """HotPathStep on models MMSSL.forward does NOT route through the packed hot node (embed_size 128, a third modality;
    a 20-wide text feature was one of them until the split-precision projection took any width % 4): the fused-AdamW hand-off of the projection weights only exists in that node, so the step
    must fall back to the optimiser launch for them - four steps (eager and captured) against the oracle stepped by
    torch.optim.AdamW: losses 1e-4, every trained tensor moved and within 5e-4 of the oracle's."""
    from mmssl_amd.graph import GraphPlan
    from mmssl_amd.hotpath import HotPathStep
    from mmssl_amd.Models import MMSSL
    from mmssl_amd import config
    U, I, dv, dt, B = 1500, 900, 96, (20 if kind == "narrow_text" else 64), 256
    d = 128 if kind == "d128" else 64
    rng = np.random.default_rng(3)
    raw = sp.csr_matrix((np.ones(12000, np.float32), (rng.integers(0, U, 12000), rng.integers(0, I, 12000))), shape=(U, I))
    raw.data[:] = 1.0
    ui, iu = O.csr_norm(raw, True).tocsr(), O.csr_norm(raw.T, True).tocsr()
    config.configure([], drop_rate=0.2, batch_size=B, weight_size=str([d] * 2), embed_size=d, debug=True)
    g = torch.Generator().manual_seed(0)
    img, txt, aud = torch.randn(I, dv, generator=g), torch.randn(I, dt, generator=g), torch.randn(I, 40, generator=g)
    extra = {"audio": aud.numpy()} if kind == "three_modalities" else None
    nmod = 3 if extra else 2
    torch.manual_seed(4)
    state0 = {k: v.detach().clone() for k, v in MMSSL(U, I, d, [d] * 2, [0.1] * 2, img.numpy(), txt.numpy(),
                                                      extra_feats=extra).state_dict().items()}
    km = [(torch.rand(I, d, generator=g) >= 0.2) for _ in range(nmod)]
    modal_m, modal_g = [], []
    for s in range(nmod):
        us = rng.choice(U, 300, replace=False)
        m = sp.csr_matrix((np.ones(300, np.float32), (us, rng.integers(0, I, 300))), shape=(U, I))
        m_ui, m_iu = O.csr_norm(m, True).tocsr(), O.csr_norm(m.T, True).tocsr()
        modal_m.append((O.to_torch_sparse(m_ui), O.to_torch_sparse(m_iu)))
        modal_g.append((GraphPlan(m_ui), GraphPlan(m_iu)))
    steps = 4
    batches = [(torch.from_numpy(rng.choice(U, B, replace=False)), torch.from_numpy(rng.integers(0, I, B)),
                torch.from_numpy(rng.integers(0, I, B))) for _ in range(steps)]
    names = ["image_trans.weight", "image_trans.bias", "text_trans.weight", "text_trans.bias", "user_id_embedding.weight",
             "item_id_embedding.weight", "weight_dict.w_self_attention_cat"] + (["audio_trans.weight", "audio_trans.bias"] if extra else [])
    skip = ("image_embedding", "text_embedding", "audio_embedding", "batch_norm", "encoder.", "align.")
    P = {k: v.clone().requires_grad_(k in names) for k, v in state0.items() if not k.startswith(skip)}
    opt = torch.optim.AdamW([P[k] for k in names], lr=5.5e-4)
    cfg = O.Cfg(drop_rate=0.2, n_ui_layers=2, batch_size=B, embed_size=d)
    A_ui, A_iu = O.to_torch_sparse(ui), O.to_torch_sparse(iu)
    ref_losses = []
    for users, pos, neg in batches:
        opt.zero_grad()
        o = O.forward_multi(P, [img, txt] + ([aud] if extra else []), (A_ui, A_iu), modal_m, cfg,
                            names=("image", "text") + (("audio",) if extra else ()), training=True,
                            keep_masks=[k.float() for k in km])
        loss = O.generator_loss_multi(o, users, pos, neg, I, cfg)
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
    graphs_g = (GraphPlan(ui), GraphPlan(iu), modal_g[0][0], modal_g[0][1], modal_g[1][0], modal_g[1][1])
    for mode in ("eager", "graph"):
        model = MMSSL(U, I, d, [d] * 2, [0.1] * 2, img.numpy(), txt.numpy(), extra_feats=extra)
        model.load_state_dict(state0)
        model = model.to(DEV).train()
        step = HotPathStep(model, graphs_g, B, decay=1e-5)
        # (a 20-wide feature runs ON the packed node since the split-precision projection zero-pads its slices: that case
        # now checks the fused hand-offs with a padded reduction instead)
        assert step._packed == (kind == "narrow_text") and step.fuse_adam == step._packed
        if extra:
            step.extra_graphs = {"audio": modal_g[2]}
        step.keep_masks = [k.to(torch.uint8).to(DEV) for k in km]
        if mode == "graph":
            step.set_batch(*(x.to(DEV) for x in batches[0]))
            assert step.capture(warmup=2), getattr(step, "capture_error", "")
            with torch.no_grad():
                for k, p in model.named_parameters():
                    p.copy_(state0[k])
            step.optimizer.reset_state()
        got = []
        for users, pos, neg in batches:
            step.set_batch(users.to(DEV), pos.to(DEV), neg.to(DEV))
            step.run()
            torch.cuda.synchronize()
            got.append(float(step.loss))
        np.testing.assert_allclose(got, ref_losses, rtol=1e-4, atol=0, err_msg=mode)
        for k in names:          # (get_parameter: named_parameters() lists a shared module under its first registration)
            got_k = model.get_parameter(k).detach().cpu()
            e = H.rel_err(got_k, P[k].detach())
            moved = H.rel_err(state0[k], P[k].detach())
            assert moved > 1e-4, (mode, k, "the oracle did not move it")
            assert H.rel_err(got_k, state0[k]) > 0.3 * moved, (mode, k, "not trained")
            assert e < 5e-4 and moved > 20 * e, (mode, k, e, moved)
