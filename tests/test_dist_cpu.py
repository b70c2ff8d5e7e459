"""world_size 2 / 3 gloo tests of the row-sharded hot path (mmssl_amd.dist) on CPU: the N-rank loss
and gradients must equal the single-process oracle on the same global inputs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _global_problem(modal):
    import helpers as H
    import mmssl_oracle as O
    fx = H.load("g8_gstep_%s.npz" % ("empty" if modal.startswith("empty") else "full"))
    d, raw, U, I = H.dataset()
    state = {k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("p.")}
    users, pos, neg = (torch.from_numpy(fx[k]) for k in ("users", "pos", "neg"))
    img_raw = H.modal_raw(fx, "img", U, I)
    txt_raw = H.modal_raw(fx, "txt", U, I)
    return fx, d, raw, U, I, state, users, pos, neg, img_raw, txt_raw


def _pad_text_to_slices(d, state, width=64):
    """The G8 text features are 48 wide; the grouped projection kernels take whole 32-deep slices. Zero-padding the
    feature columns AND the weight columns leaves every output, the loss and every other gradient unchanged (the padded
    weight columns get an exactly-zero gradient): GPU tests use it so that the packed node is the one that runs."""
    import numpy as np
    tf = np.asarray(d["text_feat"], dtype=np.float32)
    k = tf.shape[1]
    d2 = dict(d)
    d2["text_feat"] = np.concatenate([tf, np.zeros((tf.shape[0], width - k), np.float32)], 1)
    st = dict(state)
    w = state["text_trans.weight"]
    st["text_trans.weight"] = torch.cat([w, torch.zeros(w.shape[0], width - k, dtype=w.dtype)], 1)
    return d2, st, k


def _global_masks(I, d=64):
    g = torch.Generator().manual_seed(77)
    return [(torch.rand(I, d, generator=g) >= 0.2) for _ in range(2)]


def _local_pair(md, bk, O, m, ush, ish, scheme="gather-both", need_out=None):
    """(A_ui[U_r, :], A_iu[I_r, :]) - or, scheme item-side, (A_ui[U_r, :], A_iu[:, U_r]); scheme halo: the item-side pair on
    compact item columns (the referenced item ids are appended to `need_out`) - of the raw interactions m."""
    ui, iu = O.csr_norm(m, True), O.csr_norm(m.T, True)
    if scheme in ("item-side", "halo"):
        ui_l, iu_l = md.shard_graph(ui, ush, ish), md.shard_graph_cols(iu, ish, ush)
        if scheme == "halo":
            need, ui_l, iu_l = md.halo_graphs(ui_l, iu_l)
            need_out.append(need)
        return bk.make_graph(ui_l), bk.make_graph(iu_l)
    return bk.make_graph(md.shard_graph(ui, ush, ish)), bk.make_graph(md.shard_graph(iu, ish, ush))


def _worker(rank, world, port, modal, out_dir, fused=True, scheme="gather-both", chunks=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import mmssl_oracle as O
    from mmssl_amd import dist as md
    from oracle_backend import OracleBackend
    fx, d, raw, U, I, state, users, pos, neg, img_raw, txt_raw = _global_problem(modal)
    ush, ish = md.RowShard(U, world, rank), md.RowShard(I, world, rank)
    bk = OracleBackend()
    drop = modal.endswith("_drop")           # injected dropout masks: the packed keep layout and its backward
    modal = modal.replace("_drop", "")
    cfg = O.Cfg(drop_rate=0.2 if drop else 0.0, batch_size=48, n_ui_layers=2)
    repl = scheme.endswith("-repl")          # item-side with the feature matrices replicated on every rank
    scheme = scheme.replace("-repl", "")

    def local_pair(m):
        ui, iu = O.csr_norm(m, True), O.csr_norm(m.T, True)
        return bk.make_graph(md.shard_graph(ui, ush, ish)), bk.make_graph(md.shard_graph(iu, ish, ush))
    need = []
    ui, iu = _local_pair(md, bk, O, raw, ush, ish, scheme, need)   # the interaction graph in the scheme's form
    a, b = local_pair(img_raw)                                   # (the modal id graphs always as row blocks)
    c, e = local_pair(txt_raw)
    model = md.ShardedMMSSL(bk, cfg, ush, ish, state, d["image_feat"], d["text_feat"], scheme=scheme, chunks=chunks,
                            replicate_feats=repl).train()
    if scheme == "halo":
        model.halo = md.HaloPlan(need[0], ish, None, bk, torch.device("cpu"))
        assert model.halo.n_need <= ish.n_pad and sum(model.halo.recv_rows) == model.halo.n_need
    step = md.ShardedHotPathStep(model, (ui, iu, a, b, c, e), 48, I, modal_empty=(modal == "empty_shortcut"),
                                 optimizer=False, fused=fused)
    if scheme in ("item-side", "halo") and fused and world > 1:
        assert model.n_chunks(2) == max(chunks, 1)
    step.set_batch(users, pos, neg)
    if drop:       # (replicated features: the masks cover ALL item rows on every rank)
        rows = (lambda k: md._pad_rows(k, ish.n_pad)) if repl else ish.slice_rows
        step.keep_masks = tuple(rows(k.to(torch.uint8)) for k in _global_masks(I))
    md.COMM["log"] = []
    total = step.backward()
    kinds = [k for k, _, _ in md.COMM["log"]]
    md.COMM["log"] = None
    if repl and world > 1 and fused:      # the modal chain's X never travels: 2 of its 4 table-sized collectives are gone
        L, nc = cfg.n_ui_layers, max(chunks, 1)
        extra = 0 if modal == "empty_shortcut" else 2          # the modal id graphs' own table gathers (and their adjoints)
        want = (2 * L) * nc + nc + extra                         # (without replication: (2 L) nc + 2 nc + extra)
        assert kinds.count("all_gather") == want and kinds.count("reduce_scatter") == want, kinds
    torch.save({"loss": float(total), "ush": (ush.lo, ush.hi, ush.n), "ish": (ish.lo, ish.hi, ish.n),
                "g": {n: p.grad.clone() if p.grad is not None else None for n, p in model.named_parameters()}},
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _reference(modal):
    import mmssl_oracle as O
    drop = modal.endswith("_drop")
    modal = modal.replace("_drop", "")
    fx, d, raw, U, I, state, users, pos, neg, img_raw, txt_raw = _global_problem(modal)
    P = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    cfg = O.Cfg(drop_rate=0.2 if drop else 0.0, batch_size=48, n_ui_layers=2)
    pair = lambda m: O.graph_pair(m)     # noqa: E731
    ui, iu = pair(raw)
    a, b = pair(img_raw)
    c, e = pair(txt_raw)
    o = O.forward(P, torch.from_numpy(d["image_feat"]), torch.from_numpy(d["text_feat"]), (ui, iu, a, b, c, e), cfg,
                  training=drop, keep_masks=[k.float() for k in _global_masks(I)] if drop else None)
    mf, emb, _ = O.bpr(o[0][users], o[1][pos], o[1][neg], cfg.decay, 48)
    loss = mf + emb + O.feat_reg(o[2], o[3], o[4], o[5], I, cfg.feat_reg_decay) + cfg.cl_rate * (
        O.infonce(o[8][users], o[6][users], cfg.tau) + O.infonce(o[9][users], o[6][users], cfg.tau))
    loss.backward()
    return float(loss), P


@pytest.mark.parametrize("world,modal,fused,scheme,chunks", [
    (2, "full", True, "gather-both", 0), (3, "full", True, "gather-both", 0), (2, "empty", True, "gather-both", 0),
    (2, "empty_shortcut", True, "gather-both", 0), (2, "full", False, "gather-both", 0), (3, "full", False, "gather-both", 0),
    (3, "full_drop", True, "gather-both", 0),
    # item-side scheme (user-row blocks only; every collective of item-table size), whole and in column chunks; uneven
    # last blocks at world 3 (300 users / 200 items are not multiples of 3) and world 8
    (2, "full", True, "item-side", 1), (3, "full", True, "item-side", 2), (3, "full_drop", True, "item-side", 2),
    (2, "empty_shortcut", True, "item-side", 2), (3, "empty", True, "item-side", 1), (3, "full", False, "item-side", 0),
    (8, "full_drop", True, "item-side", 2),
    # item-side with the constant feature matrices replicated: the projected features never travel
    (2, "full", True, "item-side-repl", 1), (3, "full_drop", True, "item-side-repl", 2), (3, "empty_shortcut", True, "item-side-repl", 1),
    (8, "full_drop", True, "item-side-repl", 2),
    # halo scheme: only the item rows a rank's edges reference travel (all-to-all of row lists + selection SpMM)
    (2, "full", True, "halo", 1), (3, "full_drop", True, "halo", 2), (3, "empty_shortcut", True, "halo", 1),
    (8, "full_drop", True, "halo", 2)])
def test_sharded_step_equals_single_process(tmp_path, world, modal, fused, scheme, chunks):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, modal, str(tmp_path), fused, scheme, chunks), nprocs=world, join=True)
    ref_loss, P = _reference(modal)
    outs = [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(world)]
    import helpers as H
    for o in outs:
        assert abs(o["loss"] - ref_loss) <= 2e-5 * abs(ref_loss), (o["loss"], ref_loss)

    def rel(a, b):
        return float((a - b).abs().max() / (b.abs().max() + 1e-30))
    # replicated parameters: identical (all-reduced) gradients on every rank == global gradient
    for name, key in (("img_w", "image_trans.weight"), ("img_b", "image_trans.bias"), ("txt_w", "text_trans.weight"),
                      ("txt_b", "text_trans.bias")):
        for o in outs:
            assert rel(o["g"][name], P[key].grad) < 1e-4, name
    if modal.startswith("full"):
        for o in outs:
            assert rel(o["g"]["w_cat"], P["weight_dict.w_self_attention_cat"].grad) < 1e-4
    # sharded tables: each rank holds the gradient rows it owns
    for o in outs:
        for name, key, sh in (("E_u", "user_id_embedding.weight", o["ush"]), ("E_i", "item_id_embedding.weight", o["ish"])):
            lo, hi, n = sh
            k = max(0, min(hi, n) - lo)
            g = P[key].grad[lo:lo + k]
            if k > 0:
                assert rel(o["g"][name][:k], g) < 1e-4, name
                # every owned row against its own scale (floor: 1e-3 of the GLOBAL table's largest entry)
                den = torch.clamp(g.abs().amax(1), min=1e-3 * float(P[key].grad.abs().max()))
                assert float(((o["g"][name][:k] - g).abs().amax(1) / den).max()) < 5e-3, (name, "row-wise")
            if k < hi - lo:      # padded rows never receive gradient
                assert float(o["g"][name][k:].abs().max()) == 0.0


def _traj_worker(rank, world, port, out_dir, steps, scheme="gather-both", chunks=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import mmssl_oracle as O
    from mmssl_amd import dist as md
    from oracle_backend import OracleBackend
    fx, d, raw, U, I, state, users, pos, neg, img_raw, txt_raw = _global_problem("full")
    ush, ish = md.RowShard(U, world, rank), md.RowShard(I, world, rank)
    bk = OracleBackend()
    cfg = O.Cfg(drop_rate=0.0, batch_size=48, n_ui_layers=2)

    def local_pair(m):
        ui, iu = O.csr_norm(m, True), O.csr_norm(m.T, True)
        return bk.make_graph(md.shard_graph(ui, ush, ish)), bk.make_graph(md.shard_graph(iu, ish, ush))
    need = []
    graphs = _local_pair(md, bk, O, raw, ush, ish, scheme, need) + local_pair(img_raw) + local_pair(txt_raw)
    model = md.ShardedMMSSL(bk, cfg, ush, ish, state, d["image_feat"], d["text_feat"], scheme=scheme, chunks=chunks).train()
    if scheme == "halo":
        model.halo = md.HaloPlan(need[0], ish, None, bk, torch.device("cpu"))
    step = md.ShardedHotPathStep(model, graphs, 48, I, lr=1e-2)          # CPU: torch.optim.AdamW on the local tensors
    losses = []
    g = torch.Generator().manual_seed(5)
    for _ in range(steps):                                                # identical batches on every rank (global ids)
        step.set_batch(torch.randperm(U, generator=g)[:48], torch.randint(0, I, (48,), generator=g),
                       torch.randint(0, I, (48,), generator=g))
        losses.append(float(step.step()))
    torch.save({"losses": losses, "ush": (ush.lo, ush.hi, ush.n), "ish": (ish.lo, ish.hi, ish.n),
                "p": {n: p.detach().clone() for n, p in model.named_parameters()}}, os.path.join(out_dir, "t%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,scheme,chunks", [(2, "gather-both", 0), (3, "gather-both", 0), (3, "item-side", 2),
                                                 (3, "halo", 2)])
def test_sharded_trajectory_equals_single_process_adamw(tmp_path, world, scheme, chunks):
    """Four sharded steps WITH the optimiser (gloo, world 2 / 3): every rank's losses, its rows of the embedding tables and
    the replicated tensors follow the single-process oracle stepped by torch.optim.AdamW on the global problem - the
    persistent gradient bucket (gradients = views of it, the regulariser share in its last slot) across several steps."""
    import mmssl_oracle as O
    steps = 4
    port = _free_port()
    mp.spawn(_traj_worker, args=(world, port, str(tmp_path), steps, scheme, chunks), nprocs=world, join=True)
    fx, d, raw, U, I, state, users, pos, neg, img_raw, txt_raw = _global_problem("full")
    names = ("image_trans.weight", "image_trans.bias", "text_trans.weight", "text_trans.bias", "user_id_embedding.weight",
             "item_id_embedding.weight", "weight_dict.w_self_attention_cat")
    P = {k: v.clone().requires_grad_(k in names) for k, v in state.items()}
    opt = torch.optim.AdamW([P[k] for k in names], lr=1e-2)
    cfg = O.Cfg(drop_rate=0.0, batch_size=48, n_ui_layers=2)
    A = O.graph_pair(raw) + O.graph_pair(img_raw) + O.graph_pair(txt_raw)
    g = torch.Generator().manual_seed(5)
    ref = []
    for _ in range(steps):
        u_, p_, n_ = torch.randperm(U, generator=g)[:48], torch.randint(0, I, (48,), generator=g), torch.randint(0, I, (48,), generator=g)
        opt.zero_grad()
        o = O.forward(P, torch.from_numpy(d["image_feat"]), torch.from_numpy(d["text_feat"]), A, cfg, training=False)
        mf, emb, _ = O.bpr(o[0][u_], o[1][p_], o[1][n_], cfg.decay, 48)
        loss = mf + emb + O.feat_reg(o[2], o[3], o[4], o[5], I, cfg.feat_reg_decay) + cfg.cl_rate * (
            O.infonce(o[8][u_], o[6][u_], cfg.tau) + O.infonce(o[9][u_], o[6][u_], cfg.tau))
        loss.backward()
        opt.step()
        ref.append(float(loss))
    assert ref[-1] != ref[0]
    outs = [torch.load(os.path.join(str(tmp_path), "t%d.pt" % r)) for r in range(world)]

    def rel(a, b):
        return float((a - b).abs().max() / (b.abs().max() + 1e-30))
    for o in outs:
        np.testing.assert_allclose(o["losses"], ref, rtol=2e-5)
        for name, key in (("img_w", "image_trans.weight"), ("img_b", "image_trans.bias"), ("txt_w", "text_trans.weight"),
                          ("txt_b", "text_trans.bias"), ("w_cat", "weight_dict.w_self_attention_cat")):
            assert rel(o["p"][name], P[key].detach()) < 2e-4, name
        for name, key, sh in (("E_u", "user_id_embedding.weight", o["ush"]), ("E_i", "item_id_embedding.weight", o["ish"])):
            lo, hi, n = sh
            k = max(0, min(hi, n) - lo)
            assert rel(o["p"][name][:k], P[key].detach()[lo:lo + k]) < 2e-4, name


def test_row_shard_and_graph_slicing():
    import scipy.sparse as sp
    from mmssl_amd import dist as md
    m = sp.random(10, 7, density=0.4, random_state=1, format="csr", dtype=np.float32)
    parts = []
    for r in range(3):
        rs, cs = md.RowShard(10, 3, r), md.RowShard(7, 3, r)
        assert (rs.per, rs.n_pad, cs.per, cs.n_pad) == (4, 12, 3, 9)
        g = md.shard_graph(m, rs, cs)
        assert g.shape == (4, 9)
        parts.append(g)
    full = sp.vstack(parts).toarray()
    assert np.array_equal(full[:10, :7], m.toarray()) and not full[10:].any() and not full[:, 7:].any()
    t = torch.arange(20.).view(10, 2)
    assert torch.equal(md.RowShard(10, 3, 2).slice_rows(t), torch.cat([t[8:], torch.zeros(2, 2)]))
    # the item-side scheme's second graph: column blocks over ALL (padded) rows; side by side they are the matrix again
    cols = []
    for r in range(3):
        g = md.shard_graph_cols(m, md.RowShard(10, 3, r), md.RowShard(7, 3, r))
        assert g.shape == (12, 3)
        cols.append(g)
    full = sp.hstack(cols).toarray()
    assert np.array_equal(full[:10, :7], m.toarray()) and not full[10:].any() and not full[:, 7:].any()


@pytest.mark.parametrize("mode", ["ok", "fail"])
def test_spawn_rank_probe(mode):
    """bench.py decides on hipGraph capture for N>1 from per-rank child processes that form their own
    process group (dist.spawn_rank_probe). Checked here under the real launcher with gloo."""
    import subprocess
    import sys
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(here, "_probe_parent.py"), "parent", mode],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def _graph_worker(rank, world, port, scaling, out_dir, workload="tiny", scheme="gather-both"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from mmssl_amd import dist as md
    a = types.SimpleNamespace(workload=workload)
    ui_l, iu_l, ush, ish, U, I, E, dv, dt = md.build_sharded_graph(a, rank, world, torch.device("cpu"), scaling, scheme)
    torch.save({"ui": ui_l, "iu": iu_l, "U": U, "I": I, "E": E, "per": (ush.per, ish.per)},
               os.path.join(out_dir, "g%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,scaling,scheme", [(2, "weak", "gather-both"), (3, "weak", "gather-both"),
                                                  (2, "strong", "gather-both"), (3, "weak", "item-side"),
                                                  (2, "strong", "item-side")])
def test_sharded_graph_generation_is_one_consistent_global_graph(tmp_path, world, scaling, scheme):
    """bench.py's N>1 workloads: in weak mode every rank generates only its users' interactions and the edges
    reach the item owners through an all-to-all; the row blocks must assemble to ONE graph whose A_iu is the
    row-normalised transpose of the same interactions as A_ui (main.py:65-67), with nothing lost or duplicated."""
    import scipy.sparse as sp
    port = _free_port()
    mp.spawn(_graph_worker, args=(world, port, scaling, str(tmp_path), "tiny", scheme), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "g%d.pt" % r), weights_only=False) for r in range(world)]
    U, I, E = outs[0]["U"], outs[0]["I"], outs[0]["E"]
    A_ui = sp.vstack([o["ui"] for o in outs]).tocsr()[:U, :I]
    if scheme == "item-side":        # column blocks A_iu[:, U_r]: no edge left its rank, only the item degrees were summed
        for o in outs:
            assert o["iu"].shape == (o["per"][1] * world, o["per"][0])
            assert ((o["iu"] != 0).astype(np.int8) != (o["ui"] != 0).astype(np.int8).T).nnz == 0      # the rank's own edges
        A_iu = sp.hstack([o["iu"] for o in outs]).tocsr()[:I, :U]
    else:
        A_iu = sp.vstack([o["iu"] for o in outs]).tocsr()[:I, :U]
    assert A_ui.nnz == E == A_iu.nnz
    pat = (A_ui != 0).astype(np.float32)
    assert ((A_iu != 0).astype(np.float32) != pat.T).nnz == 0            # same interactions, transposed
    from mmssl_amd import synth
    ref_ui, ref_iu = synth.normalised_pair(pat.tocsr())
    assert abs(A_ui - ref_ui).max() < 1e-6 and abs(A_iu - ref_iu).max() < 1e-6
    if scaling == "weak":
        assert (U, I) == (600 * world, 400 * world)


def test_baby_strong_scaling_partition_world8(tmp_path):
    """BASELINE configs[3]: the Amazon-Baby graph itself cut 8 ways (`bench.py --gpus 8 --scaling strong`). The eight
    row blocks reassemble to the one global graph, every rank holds ceil(n / 8) rows (the last one zero-padded), and
    the column space is padded to a multiple of 8 (what the all-gathered tables have)."""
    import scipy.sparse as sp
    from mmssl_amd import synth
    world = 8
    port = _free_port()
    mp.spawn(_graph_worker, args=(world, port, "strong", str(tmp_path), "baby"), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "g%d.pt" % r), weights_only=False) for r in range(world)]
    U, I, E, _, _ = synth.SHAPES["baby"]
    per_u, per_i = -(-U // world), -(-I // world)
    for o in outs:
        assert (o["U"], o["I"]) == (U, I) and o["per"] == (per_u, per_i)
        assert o["ui"].shape == (per_u, per_i * world) and o["iu"].shape == (per_i, per_u * world)
    raw = synth.interaction_matrix(U, I, E, seed=1)
    ref_ui, ref_iu = synth.normalised_pair(raw)
    A_ui = sp.vstack([o["ui"] for o in outs]).tocsr()
    A_iu = sp.vstack([o["iu"] for o in outs]).tocsr()
    assert A_ui[U:].nnz == 0 and A_ui[:, I:].nnz == 0 and A_iu[I:].nnz == 0 and A_iu[:, U:].nnz == 0     # padding is empty
    assert (A_ui[:U, :I] != ref_ui).nnz == 0 and (A_iu[:I, :U] != ref_iu).nnz == 0                          # bit-identical values
    assert outs[0]["E"] == raw.nnz == sum(o["ui"].nnz for o in outs) == sum(o["iu"].nnz for o in outs)


def test_replicate_feats_choice_and_row_padding():
    """dist.choose_replicate_feats: narrow features on many items (configs[4]) -> replicate; wide features (the Baby shape
    x 8) -> ship the projected ones; never at world 1. dist._pad_rows keeps the rows and zero-fills the padding."""
    from mmssl_amd import dist as md
    assert md.choose_replicate_feats(1_000_000, [128, 128], 128, 8)
    assert md.choose_replicate_feats(250_000, [128, 128], 128, 2)
    assert not md.choose_replicate_feats(18357 * 8, [4096, 1024], 64, 8)
    assert not md.choose_replicate_feats(1_000_000, [128, 128], 128, 1)
    t = torch.arange(12.0).reshape(4, 3)
    p = md._pad_rows(t, 6)
    assert p.shape == (6, 3) and torch.equal(p[:4], t) and float(p[4:].abs().max()) == 0.0
    assert md._pad_rows(t, 4).data_ptr() != t.data_ptr()
    with pytest.raises(ValueError):
        md.ShardedMMSSL(object(), None, md.RowShard(4, 1, 0), md.RowShard(4, 1, 0), {}, t, t, scheme="gather-both",
                        replicate_feats=True)
