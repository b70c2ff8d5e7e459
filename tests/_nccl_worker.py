"""Child process of tests/test_dist_nccl_gpu.py: the row-sharded hot path on the PRODUCT backend with a real RCCL
process group (backend "nccl", world size 1) and MMSSL_DIST_FORCE_COLLECTIVES=1, so that every collective of the
N > 1 code path (all_gather_into_tensor, reduce_scatter_tensor + the fused add, the flat gradient all-reduce, the
batch-row all-reduce, RCCL launches inside a hipGraph capture) really executes on the GPU. Runs in its own process
because a process has one default group (the gloo world-1 tests own the pytest process's) and because a failed capture
with RCCL inside can abort the process.

    python tests/_nccl_worker.py CASE OUT.json        CASE in {g8, baby, synth_rank}
"""
import json
import os
import socket
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _init():
    import torch.distributed as dist
    os.environ["MMSSL_DIST_FORCE_COLLECTIVES"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    return dist


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


import helpers as H   # noqa: E402  (tests/ is on sys.path above)

_row_rel = H.row_rel
# which sharding scheme / how many column chunks the cases run (the parent sets them per test)
SCHEME = os.environ.get("MMSSL_TEST_SCHEME", "gather-both")
CHUNKS = int(os.environ.get("MMSSL_TEST_CHUNKS", "0"))


def _iu_local(md, iu, ish, ush):
    return md.shard_graph_cols(iu, ish, ush) if SCHEME in ("item-side", "halo") else md.shard_graph(iu, ish, ush)


def _halo(md, model, need, ish, bk, dev):
    if SCHEME == "halo":
        model.halo = md.HaloPlan(need, ish, None, bk, dev)


def _comm_kinds(md, fn):
    md.COMM["log"] = []
    fn()
    log, md.COMM["log"] = md.COMM["log"], None
    return {k: sum(1 for x in log if x[0] == k) for k in ("all_gather", "reduce_scatter", "all_reduce")}


def case_g8(out):
    """G8 problem (reference-recorded parameters and batch): eager and captured, full and empty modal graphs."""
    import mmssl_oracle as O
    import test_dist_cpu as T
    from mmssl_amd import dist as md
    dev = torch.device("cuda", 0)
    for modal in ("full", "empty_shortcut"):
        fx, d, raw, U, I, state, users, pos, neg, img_raw, txt_raw = T._global_problem(modal)
        ush, ish = md.RowShard(U, 1, 0), md.RowShard(I, 1, 0)
        bk = md.HipBackend()
        cfg = O.Cfg(drop_rate=0.0, batch_size=48, n_ui_layers=2)

        def local_pair(m):
            ui, iu = O.csr_norm(m, True), O.csr_norm(m.T, True)
            return bk.make_graph(md.shard_graph(ui, ush, ish)), bk.make_graph(md.shard_graph(iu, ish, ush))
        need = []
        graphs = T._local_pair(md, bk, O, raw, ush, ish, SCHEME, need) + local_pair(img_raw) + local_pair(txt_raw)
        d, state, k_txt = T._pad_text_to_slices(d, state)      # whole 32-deep slices: the packed node runs
        model = md.ShardedMMSSL(bk, cfg, ush, ish, state, d["image_feat"], d["text_feat"], scheme=SCHEME,
                                chunks=CHUNKS).to(dev).train()
        _halo(md, model, need[0] if need else None, ish, bk, dev)
        step = md.ShardedHotPathStep(model, graphs, 48, I, modal_empty=(modal == "empty_shortcut"), optimizer=False)
        step.set_batch(torch.stack([users, pos, neg]).to(dev))
        ref_loss, P = T._reference(modal)
        names = [("img_w", "image_trans.weight"), ("img_b", "image_trans.bias"), ("txt_w", "text_trans.weight"),
                 ("txt_b", "text_trans.bias"), ("E_u", "user_id_embedding.weight"), ("E_i", "item_id_embedding.weight")]
        if modal == "full":
            names.append(("w_cat", "weight_dict.w_self_attention_cat"))

        def check(tag):
            torch.cuda.synchronize()
            g = {n: p.grad for n, p in model.named_parameters()}
            assert model.last_fused and float(g["txt_w"][:, k_txt:].abs().max()) == 0.0
            g["txt_w"] = g["txt_w"][:, :k_txt]
            rec = {"loss_rel": abs(float(step.loss) - ref_loss) / abs(ref_loss)}
            for name, key in names:
                k = P[key].grad.shape[0]
                rec[name] = _rel(g[name][:k], P[key].grad)
            out["g8/%s/%s" % (modal, tag)] = rec
        out["g8/%s/collectives" % modal] = _comm_kinds(md, step.step)
        check("eager")
        ok = step.capture(warmup=2)
        out["g8/%s/captured" % modal] = bool(ok)
        if ok:
            for p in model.parameters():
                if p.grad is not None:
                    p.grad.zero_()
            step.run()
            check("replay")
        else:
            out["g8/%s/capture_error" % modal] = getattr(step, "capture_error", "")
    # trajectory with the optimiser inside the captured step: 4 replays == 4 eager steps
    fx, d, raw, U, I, state, users, pos, neg, img_raw, txt_raw = T._global_problem("full")
    losses = {}
    for mode in ("eager", "graph"):
        ush, ish = md.RowShard(U, 1, 0), md.RowShard(I, 1, 0)
        bk = md.HipBackend()
        cfg = O.Cfg(drop_rate=0.0, batch_size=48, n_ui_layers=2)
        need = []
        graphs = T._local_pair(md, bk, O, raw, ush, ish, SCHEME, need)
        for m in (img_raw, txt_raw):
            ui, iu = O.csr_norm(m, True), O.csr_norm(m.T, True)
            graphs += (bk.make_graph(md.shard_graph(ui, ush, ish)), bk.make_graph(md.shard_graph(iu, ish, ush)))
        d2, st2, _ = T._pad_text_to_slices(d, state)
        model = md.ShardedMMSSL(bk, cfg, ush, ish, st2, d2["image_feat"], d2["text_feat"], scheme=SCHEME,
                                chunks=CHUNKS).to(dev).train()
        _halo(md, model, need[0] if need else None, ish, bk, dev)
        step = md.ShardedHotPathStep(model, graphs, 48, I, lr=1e-2)
        step.set_batch(torch.stack([users, pos, neg]).to(dev))
        snap = [p.detach().clone() for p in model.parameters()]
        if mode == "graph":
            assert step.capture(warmup=2), getattr(step, "capture_error", "")
            with torch.no_grad():
                for p, q in zip(model.parameters(), snap):
                    p.copy_(q)
            step.optimizer.reset_state()
        ls = []
        for _ in range(4):
            step.run()
            torch.cuda.synchronize()
            ls.append(float(step.loss))
        losses[mode] = ls
    out["g8/trajectory"] = losses


def _baby_problem(dev, ref=True):
    """`ref=False`: inputs only (no oracle forward / backward)."""
    import mmssl_oracle as O
    from mmssl_amd import config, synth, dist as md
    from mmssl_amd.Models import MMSSL
    U, I, E, dv, dt = synth.SHAPES["baby"]
    config.configure([], drop_rate=0.2, batch_size=1024, weight_size=str([64] * 3), debug=True)
    a = types.SimpleNamespace(workload="baby")
    ui_l, iu_l, ush, ish, U, I, E, dv, dt = md.build_sharded_graph(a, 0, 1, dev, "strong", "item-side" if SCHEME == "halo" else SCHEME)
    g = torch.Generator().manual_seed(0)
    img, txt = torch.randn(I, dv, generator=g), torch.randn(I, dt, generator=g)
    torch.manual_seed(4)
    cpu_model = MMSSL(U, I, 64, [64] * 3, [0.1] * 3, img.numpy(), txt.numpy())
    state = {k: v.detach().clone() for k, v in cpu_model.state_dict().items()
             if not k.startswith(("image_embedding", "text_embedding", "batch_norm", "encoder.", "align."))}
    del cpu_model
    km = [(torch.rand(I, 64, generator=g) >= 0.2) for _ in range(2)]
    rng = np.random.default_rng(1)
    users = torch.from_numpy(rng.choice(U, 1024, replace=False))
    pos = torch.from_numpy(rng.integers(0, I, 1024))
    neg = torch.from_numpy(rng.integers(0, I, 1024))
    import scipy.sparse as sp
    e_ui, e_iu = sp.csr_matrix((U, I), dtype=np.float32), sp.csr_matrix((I, U), dtype=np.float32)
    A = [O.to_torch_sparse(x) for x in (ui_l[:U, :I], iu_l[:I, :U], e_ui, e_iu, e_ui, e_iu)]
    P = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in state.items()}
    cfg = O.Cfg(drop_rate=0.2, n_ui_layers=3, batch_size=1024)
    loss = float("nan")
    if ref:
        o = O.forward(P, img, txt, A, cfg, training=True, keep_masks=[k.float() for k in km])
        mf, emb, _ = O.bpr(o[0][users], o[1][pos], o[1][neg], 1e-5, 1024)
        tot = mf + emb + O.feat_reg(o[2], o[3], o[4], o[5], I, 1e-5) + 0.03 * (
            O.infonce(o[8][users], o[6][users], 0.5) + O.infonce(o[9][users], o[6][users], 0.5))
        tot.backward()
        loss = float(tot)
    return dict(U=U, I=I, ui=ui_l, iu=iu_l, ush=ush, ish=ish, state=state, img=img, txt=txt, km=km,
                batch=torch.stack([users, pos, neg]), ref=loss, P=P, e_ui=e_ui, e_iu=e_iu, cfg=cfg)


def case_baby(out):
    """configs[3]'s graph (the Amazon-Baby shape through the sharded step; at world 1 the shard is the whole graph):
    eager and captured against the oracle's step on the same parameters, dropout masks and batch."""
    from mmssl_amd import dist as md
    dev = torch.device("cuda", 0)
    pb = _baby_problem(dev)
    bk = md.HipBackend()
    ui_l, iu_l, need = pb["ui"], pb["iu"], None
    if SCHEME == "halo":
        need, ui_l, iu_l = md.halo_graphs(ui_l, iu_l)
    plans = (bk.make_graph(ui_l), bk.make_graph(iu_l), bk.make_graph(pb["e_ui"]), bk.make_graph(pb["e_iu"]))
    graphs = (plans[0], plans[1], plans[2], plans[3], plans[2], plans[3])
    model = md.ShardedMMSSL(bk, pb["cfg"], pb["ush"], pb["ish"], pb["state"], pb["img"].numpy(), pb["txt"].numpy(),
                            scheme=SCHEME, chunks=CHUNKS)
    model = model.to(dev).train()
    _halo(md, model, need, pb["ish"], bk, dev)
    out["baby/chunks"] = model.n_chunks(2) if SCHEME in ("item-side", "halo") else 1
    step = md.ShardedHotPathStep(model, graphs, 1024, pb["I"], modal_empty=True, optimizer=False)
    step.keep_masks = tuple(k.to(torch.uint8).to(dev) for k in pb["km"])
    step.set_batch(pb["batch"].to(dev))
    names = [("img_w", "image_trans.weight"), ("img_b", "image_trans.bias"), ("txt_w", "text_trans.weight"),
             ("txt_b", "text_trans.bias"), ("E_u", "user_id_embedding.weight"), ("E_i", "item_id_embedding.weight")]

    def check(tag):
        torch.cuda.synchronize()
        g = {n: p.grad for n, p in model.named_parameters()}
        rec = {"loss_rel": abs(float(step.loss) - pb["ref"]) / abs(pb["ref"])}
        for name, key in names:
            k = pb["P"][key].grad.shape[0]
            rec[name] = _rel(g[name][:k], pb["P"][key].grad)
            rec[name + "_rowwise"] = _row_rel(g[name][:k], pb["P"][key].grad)
        out["baby/" + tag] = rec
    out["baby/collectives"] = _comm_kinds(md, step.step)
    check("eager")
    ok = step.capture(warmup=2)
    out["baby/captured"] = bool(ok)
    if ok:
        for p in model.parameters():
            if p.grad is not None:
                p.grad.zero_()
        step.run()
        check("replay")
    else:
        out["baby/capture_error"] = getattr(step, "capture_error", "")


def case_synth_rank(out):
    """configs[4]'s per-rank share (250 K users x 125 K items x 12.5 M edges, d = 128): the SpMM on sampled rows against
    the oracle, adjointness of the transposed launch, bitwise determinism, and two sharded steps (forced collectives)
    with a finite, moving loss."""
    import mmssl_oracle as O
    from mmssl_amd import dist as md, ops
    dev = torch.device("cuda", 0)
    a = types.SimpleNamespace(workload="synth", d=128, gcn_layers=3, batch=1024, scheme=SCHEME, chunks=CHUNKS)
    step, (ui_l, iu_l), plans, stats = md.build_bench_step(a, 0, 1, dev, "weak")
    out["synth/scheme"] = [stats["scheme"], stats["chunks"]]
    out["synth/shape"] = {k: stats[k] for k in ("local_users", "local_items", "local_edges", "spmm_launches")}
    g = torch.Generator().manual_seed(0)
    for name, plan, mat in (("ui", plans[0], ui_l), ("iu", plans[1], iu_l)):
        X = torch.randn(mat.shape[1], 128, generator=g)
        Yv = torch.randn(mat.shape[0], 128, generator=g)
        Xd, Yd = X.to(dev), Yv.to(dev)
        Y = ops.spmm(plan, Xd)
        rows = np.sort(np.random.default_rng(3).choice(mat.shape[0], 3000, replace=False))
        heavy = np.argsort(np.diff(mat.indptr))[-8:]                    # the longest rows (multi-block path)
        rows = np.unique(np.concatenate([rows, heavy]))
        ref = O.spmm(O.to_torch_sparse(mat[rows]), X)
        rec = {"rows_vs_oracle": _rel(Y[torch.from_numpy(rows).to(dev)], ref),
               "max_row_nnz": int(np.diff(mat.indptr).max())}
        YT = ops.spmm(plan, Yd, transpose=True)
        lhs = float((Y.double() * Yd.double()).sum())
        rhs = float((Xd.double() * YT.double()).sum())
        rec["adjoint_rel"] = abs(lhs - rhs) / abs(lhs)
        rec["deterministic"] = bool(torch.equal(Y, ops.spmm(plan, Xd)) and torch.equal(YT, ops.spmm(plan, Yd, transpose=True)))
        cols = np.sort(np.random.default_rng(4).choice(mat.shape[1], 2000, replace=False))
        refT = O.spmm(O.to_torch_sparse(mat[:, cols].T.tocsr()), Yv)
        rec["transpose_rows_vs_oracle"] = _rel(YT[torch.from_numpy(cols).to(dev)], refT)
        out["synth/spmm_" + name] = rec
        del X, Yv, Xd, Yd, Y, YT
    rngb = np.random.default_rng(2022)
    U, I = stats["n_users"], stats["n_items"]
    losses = []
    for _ in range(2):
        b = torch.stack([torch.from_numpy(x) for x in (rngb.choice(U, 1024, replace=False).astype(np.int64),
                                                       rngb.integers(0, I, 1024).astype(np.int64),
                                                       rngb.integers(0, I, 1024).astype(np.int64))])
        step.set_batch(b.to(dev))
        step.step()
        torch.cuda.synchronize()
        losses.append(float(step.loss))
    out["synth/losses"] = losses
    out["synth/grads_finite"] = bool(all(torch.isfinite(p.grad).all().item() for p in step.model.parameters()
                                         if p.grad is not None))


def main():
    import faulthandler
    faulthandler.enable()
    case, path = sys.argv[1], sys.argv[2]
    dist = _init()
    out = {"case": case, "backend": dist.get_backend()}
    try:
        {"g8": case_g8, "baby": case_baby, "synth_rank": case_synth_rank}[case](out)
        out["ok"] = True
    except Exception as e:                                  # the parent prints this
        import traceback
        out["ok"], out["error"] = False, traceback.format_exc()
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    torch.cuda.synchronize()
    dist.destroy_process_group()
    sys.exit(0 if out["ok"] else 4)


if __name__ == "__main__":
    main()
