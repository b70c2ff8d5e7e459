"""The row-sharded hot path on the PRODUCT backend (dist.HipBackend: HIP kernels incl. the owned-row gather / scatter
pair) at world size 1 against the single-process oracle on the G8 problem: loss and every gradient, fused and composed
paths. (World sizes 2 and 3 run on CPU with the oracle backend in test_dist_cpu.py; N > 1 on GPUs is the driver's.)"""
import os
import sys

import pytest
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import helpers as H   # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solo_group():
    import test_dist_cpu as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(T._free_port())
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    yield
    if created:
        dist.destroy_process_group()


@pytest.mark.parametrize("modal,fused", [("full", True), ("empty", True), ("empty_shortcut", True), ("full", False)])
def test_sharded_step_on_hip_backend_world1_equals_oracle(solo_group, modal, fused):
    import mmssl_oracle as O
    import test_dist_cpu as T
    from mmssl_amd import dist as md
    dev = torch.device("cuda")
    fx, d, raw, U, I, state, users, pos, neg, img_raw, txt_raw = T._global_problem(modal)
    ush, ish = md.RowShard(U, 1, 0), md.RowShard(I, 1, 0)
    bk = md.HipBackend()
    cfg = O.Cfg(drop_rate=0.0, batch_size=48, n_ui_layers=2)

    def local_pair(m):
        ui, iu = O.csr_norm(m, True), O.csr_norm(m.T, True)
        return bk.make_graph(md.shard_graph(ui, ush, ish)), bk.make_graph(md.shard_graph(iu, ish, ush))
    graphs = local_pair(raw) + local_pair(img_raw) + local_pair(txt_raw)
    d, state, k_txt = T._pad_text_to_slices(d, state)
    model = md.ShardedMMSSL(bk, cfg, ush, ish, state, d["image_feat"], d["text_feat"]).to(dev).train()
    step = md.ShardedHotPathStep(model, graphs, 48, I, modal_empty=(modal == "empty_shortcut"), optimizer=False,
                                 fused=fused)
    step.set_batch(torch.stack([users, pos, neg]).to(dev))           # packed form: one copy
    total = step.backward()
    torch.cuda.synchronize()
    ref_loss, P = T._reference(modal)
    assert abs(float(total) - ref_loss) <= 2e-5 * abs(ref_loss), (float(total), ref_loss)

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-30))
    assert model.last_fused == fused                 # the packed node really ran
    g = {n: p.grad for n, p in model.named_parameters()}
    assert float(g["txt_w"][:, k_txt:].abs().max()) == 0.0
    g["txt_w"] = g["txt_w"][:, :k_txt]
    for name, key in (("img_w", "image_trans.weight"), ("img_b", "image_trans.bias"), ("txt_w", "text_trans.weight"),
                      ("txt_b", "text_trans.bias"), ("E_u", "user_id_embedding.weight"), ("E_i", "item_id_embedding.weight")):
        k = P[key].grad.shape[0]
        H.check_grad(g[name][:k], P[key].grad, 1e-4, name)
    if modal == "full":
        assert rel(g["w_cat"], P["weight_dict.w_self_attention_cat"].grad) < 1e-4


@pytest.mark.parametrize("forced", [False, True])
def test_sharded_packed_node_at_d128_per_modality_projection_equals_oracle(solo_group, forced, monkeypatch):
    """configs[4]'s width (d = 128): the grouped projection kernels are 64 channels wide, so the packed node runs the
    projection per modality (csrc/linear.hip) around the same packed d = 256 modal chain and two-sided fuse kernels. One
    sharded step (empty modal graphs, injected dropout masks) against the oracle; `forced`: the N > 1 branches (separate
    reduce-scatter adds, mask after the scatter) with identity collectives on the gloo group."""
    import scipy.sparse as sp
    import numpy as np
    import mmssl_oracle as O
    from mmssl_amd import dist as md
    if forced:
        monkeypatch.setenv("MMSSL_DIST_FORCE_COLLECTIVES", "1")
    dev = torch.device("cuda")
    U, I, d, dv, dt, B = 700, 420, 128, 96, 40, 64
    g = torch.Generator().manual_seed(8)
    rng = np.random.default_rng(8)
    raw = sp.csr_matrix((np.ones(4000, np.float32), (rng.integers(0, U, 4000), rng.integers(0, I, 4000))), shape=(U, I))
    raw.data[:] = 1.0
    img, txt = torch.randn(I, dv, generator=g), torch.randn(I, dt, generator=g)
    state = {"image_trans.weight": torch.randn(d, dv, generator=g) / dv ** 0.5, "image_trans.bias": torch.randn(d, generator=g) * 0.1,
             "text_trans.weight": torch.randn(d, dt, generator=g) / dt ** 0.5, "text_trans.bias": torch.randn(d, generator=g) * 0.1,
             "user_id_embedding.weight": torch.randn(U, d, generator=g) * 0.1,
             "item_id_embedding.weight": torch.randn(I, d, generator=g) * 0.1,
             "weight_dict.w_q": torch.randn(d, d, generator=g) * 0.05, "weight_dict.w_k": torch.randn(d, d, generator=g) * 0.05,
             "weight_dict.w_self_attention_cat": torch.randn(4 * d, d, generator=g) * 0.05}
    cfg = O.Cfg(embed_size=d, drop_rate=0.2, batch_size=B, n_ui_layers=2)
    km = [(torch.rand(I, d, generator=g) >= 0.2) for _ in range(2)]
    users = torch.randperm(U, generator=g)[:B]
    pos, neg = torch.randint(0, I, (B,), generator=g), torch.randint(0, I, (B,), generator=g)
    empty = sp.csr_matrix((U, I), dtype=np.float32)
    # oracle
    P = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    A = [O.to_torch_sparse(O.csr_norm(raw, True)), O.to_torch_sparse(O.csr_norm(raw.T, True))]
    E = [O.to_torch_sparse(empty), O.to_torch_sparse(empty.T.tocsr())]
    o = O.forward(P, img, txt, (A[0], A[1], E[0], E[1], E[0], E[1]), cfg, training=True, keep_masks=[k.float() for k in km])
    mf, emb, _ = O.bpr(o[0][users], o[1][pos], o[1][neg], cfg.decay, B)
    ref = mf + emb + O.feat_reg(o[2], o[3], o[4], o[5], I, cfg.feat_reg_decay) + cfg.cl_rate * (
        O.infonce(o[8][users], o[6][users], cfg.tau) + O.infonce(o[9][users], o[6][users], cfg.tau))
    ref.backward()
    # sharded step, one rank
    bk = md.HipBackend()
    ush, ish = md.RowShard(U, 1, 0), md.RowShard(I, 1, 0)
    graphs = tuple(bk.make_graph(m) for m in (O.csr_norm(raw, True), O.csr_norm(raw.T, True), empty, empty.T.tocsr(),
                                              empty, empty.T.tocsr()))
    model = md.ShardedMMSSL(bk, cfg, ush, ish, state, img.numpy(), txt.numpy()).to(dev).train()
    step = md.ShardedHotPathStep(model, graphs, B, I, modal_empty=True, optimizer=False)
    step.set_batch(torch.stack([users, pos, neg]).to(dev))
    step.keep_masks = tuple(k.to(torch.uint8).to(dev) for k in km)
    total = step.backward()
    torch.cuda.synchronize()
    assert model.last_fused and not bk._grouped([dv, dt], I, d)          # packed node, per-modality projection
    assert abs(float(total) - float(ref)) <= 2e-5 * abs(float(ref)), (float(total), float(ref))

    def rel(a, b):
        return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-30))
    gp = {n: p.grad for n, p in model.named_parameters()}
    for name, key in (("img_w", "image_trans.weight"), ("img_b", "image_trans.bias"), ("txt_w", "text_trans.weight"),
                      ("txt_b", "text_trans.bias"), ("E_u", "user_id_embedding.weight"), ("E_i", "item_id_embedding.weight")):
        H.check_grad(gp[name], P[key].grad, 1e-4, name)


@pytest.mark.parametrize("world,modal,scheme,chunks", [(2, "full", "item-side", 2), (3, "full_drop", "item-side", 2),
                                                       (3, "empty_shortcut", "item-side", 1), (3, "full", "gather-both", 0),
                                                       (3, "baby", "item-side", 2), (8, "baby", "item-side", 1),
                                                       (8, "baby", "gather-both", 0), (3, "full_drop", "halo", 2),
                                                       (8, "baby", "halo", 1), (3, "full_drop", "item-side-repl", 2),
                                                       (8, "baby", "item-side-repl", 1)])
def test_hip_backend_at_world_2_3_8_on_one_gpu(tmp_path, world, modal, scheme, chunks):
    _world_on_one_gpu(tmp_path, world, modal, scheme, chunks, None)


@pytest.mark.parametrize("world,modal,scheme,chunks", [(2, "full", "item-side", 2), (3, "full_drop", "item-side", 2),
                                                       (3, "empty_shortcut", "item-side", 1), (3, "full", "gather-both", 0),
                                                       (3, "baby", "item-side", 2), (8, "baby", "item-side", 1),
                                                       (3, "full_drop", "item-side-repl", 2), (8, "baby", "item-side-repl", 1)])
def test_peer_exchange_at_world_2_3_8_on_one_gpu(tmp_path, world, modal, scheme, chunks):
    """The same sharded steps with the PEER EXCHANGE as the transport (csrc/peer.hip: every all-gather is a kernel that
    pushes the rank's rows into all peers' IPC-mapped windows, every reduce-scatter a kernel that pulls the rank's rows out
    of them in rank order, the small all-reduces likewise): no torch.distributed call carries a device tensor (counted in
    the worker), no device-wide fence, the lanes' overlap intact. Loss and gradients against the single-process oracle as
    above; a second step on the same inputs reproduces the first (windows reused behind the step barrier)."""
    recs = _world_on_one_gpu(tmp_path, world, modal, scheme, chunks, "peer")
    for o in recs:
        assert o["peer"]["world"] == world and o["peer"]["call_sites"] >= 6 and o["peer"]["launches"] > 0, o["peer"]


def _world_on_one_gpu(tmp_path, world, modal, scheme, chunks, transport):
    """dist.HipBackend at world size > 1: `world` processes share GPU 0 (their group is gloo - RCCL refuses two ranks on one
    device - moving DEVICE tensors), each runs its shard of the sharded step on the HIP kernels: real per-rank partial
    products, uneven last blocks (300 users / 200 items over 3 ranks), lanes on real streams with row-pitched column-chunk
    SpMMs, the batch-rows fuse on owned rows only. Loss and every gradient against the single-process oracle, the sharded
    tables row by row. The (8, "baby", ...) cases are BASELINE configs[3]'s partition itself: the Amazon-Baby graph, item and
    user rows sharded 8 ways (4450 / 2295 rows per rank, the last blocks short), both schemes."""
    import subprocess
    import test_dist_cpu as T
    port = T._free_port()
    env = dict(os.environ)
    if transport:
        env["MMSSL_TEST_TRANSPORT"] = transport
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_hip_world_worker.py"), str(r), str(world), str(port), modal,
                               scheme, str(chunks), str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                              env=env)
             for r in range(world)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    if modal == "baby":            # configs[3]: the Baby graph cut 3 ways; reference = the oracle's step (north_star: loss 1e-4)
        import _nccl_worker as W
        pb = W._baby_problem(torch.device("cpu"), ref=True)
        ref_loss, P, tol_l, tol_g = pb["ref"], pb["P"], 1e-4, 5e-4
    else:
        (ref_loss, P), tol_l, tol_g = T._reference(modal), 2e-5, 1e-4
    recs = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(world)]
    for o in recs:
        assert abs(o["loss"] - ref_loss) <= tol_l * abs(ref_loss), (o["loss"], ref_loss)
        assert o["chunks"] == max(chunks, 1)
        for name, key in (("img_w", "image_trans.weight"), ("img_b", "image_trans.bias"), ("txt_w", "text_trans.weight"),
                          ("txt_b", "text_trans.bias")):
            H.check_grad(o["g"][name], P[key].grad, tol_g, name)
        if modal.startswith("full"):
            H.check_grad(o["g"]["w_cat"], P["weight_dict.w_self_attention_cat"].grad, tol_g, "w_cat")
        for name, key, sh in (("E_u", "user_id_embedding.weight", o["ush"]), ("E_i", "item_id_embedding.weight", o["ish"])):
            lo, hi, n = sh
            k = max(0, min(hi, n) - lo)
            if k > 0:
                ref = P[key].grad[lo:lo + k]
                assert H.rel_err(o["g"][name][:k], ref) < tol_g, name
                den = torch.clamp(ref.abs().amax(1), min=1e-3 * float(P[key].grad.abs().max()))
                assert float(((o["g"][name][:k] - ref).abs().amax(1) / den).max()) < 5e-3, (name, "row-wise")
            if k < hi - lo:
                assert float(o["g"][name][k:].abs().max()) == 0.0
    return recs
