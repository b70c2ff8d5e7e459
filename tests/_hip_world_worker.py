"""Child rank of tests/test_dist_gpu.py::test_hip_backend_at_world_2_3_8_on_one_gpu: the row-sharded hot path on the
PRODUCT backend (dist.HipBackend: HIP kernels, lanes on real streams, row-pitched column-chunk SpMMs) at world size > 1 with
every rank on the SAME GPU. RCCL refuses two ranks on one device, so the ranks talk through a gloo group (device tensors:
gloo's device all-reduce; all-gather as an all-reduce of a zero-padded buffer) - different transport, the same shards, the
same per-rank partial products, uneven last blocks and all.

    python tests/_hip_world_worker.py RANK WORLD PORT MODAL SCHEME CHUNKS OUT_DIR"""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "oracle"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


DEVICE_COLLECTIVES = {"n": 0}


def _count_device_collectives():
    for name in ("all_reduce", "all_gather_into_tensor", "reduce_scatter_tensor", "all_to_all_single", "all_gather",
                 "broadcast"):
        inner = getattr(dist, name)

        def wrapped(*a, _inner=inner, **k):
            if any(torch.is_tensor(x) and x.is_cuda for x in list(a) + list(k.values())):
                DEVICE_COLLECTIVES["n"] += 1
            return _inner(*a, **k)
        setattr(dist, name, wrapped)


def _finish_peer(md, step, total, model, rec):
    """Peer transport only: a SECOND step on the same inputs (every window is reused: the begin-of-step barrier and the
    epochs are what keep it right) must give the same result; no wait timed out; no collective carried a device tensor."""
    pc = md._peer(None)
    if pc is None:
        return rec
    g1 = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    l1 = float(total)
    sites = pc.stats()["call_sites"]
    total2 = step.backward()
    torch.cuda.synchronize()
    pc.check()
    assert abs(float(total2) - l1) <= 1e-6 * abs(l1), (float(total2), l1)
    for n, p in model.named_parameters():
        if p.grad is not None:        # (the batch rows' scatter uses fp32 atomics: equal up to their summation order)
            assert float((p.grad - g1[n]).abs().max()) <= 2e-5 * float(g1[n].abs().max()) + 1e-30, n
    st = pc.stats()
    assert st["call_sites"] == sites and DEVICE_COLLECTIVES["n"] == 0, (st, sites, DEVICE_COLLECTIVES)
    rec["peer"] = st
    return rec


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    modal, scheme, chunks, out_dir = sys.argv[4], sys.argv[5], int(sys.argv[6]), sys.argv[7]
    repl = scheme.endswith("-repl")              # item-side with the constant feature matrices on every rank
    scheme = scheme.replace("-repl", "")
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mmssl_oracle as O
    import test_dist_cpu as T
    from mmssl_amd import dist as md
    dev = torch.device("cuda", 0)
    if os.environ.get("MMSSL_TEST_TRANSPORT") == "peer":
        # the exchanges go through IPC-mapped windows + epoch flags (csrc/peer.hip): from here on NO torch.distributed
        # call may carry a device tensor - counted, and asserted to be zero when the rank is done
        md.enable_peer_exchange(None, dev, timeout_ms=120000)
        _count_device_collectives()
    if modal == "baby":
        return baby(rank, world, scheme, chunks, out_dir, md, dev, repl)
    if modal == "synth_full":
        return synth_full(rank, world, scheme, chunks, out_dir, md, dev)
    fx, d, raw, U, I, state, users, pos, neg, img_raw, txt_raw = T._global_problem(modal)
    ush, ish = md.RowShard(U, world, rank), md.RowShard(I, world, rank)
    bk = md.HipBackend()
    drop = modal.endswith("_drop")
    cfg = O.Cfg(drop_rate=0.2 if drop else 0.0, batch_size=48, n_ui_layers=2)
    d, state, k_txt = T._pad_text_to_slices(d, state)            # whole 32-deep slices: the packed node runs

    def row_pair(m):
        ui, iu = O.csr_norm(m, True), O.csr_norm(m.T, True)
        return bk.make_graph(md.shard_graph(ui, ush, ish)), bk.make_graph(md.shard_graph(iu, ish, ush))
    need = []
    graphs = T._local_pair(md, bk, O, raw, ush, ish, scheme, need) + row_pair(img_raw) + row_pair(txt_raw)
    model = md.ShardedMMSSL(bk, cfg, ush, ish, state, d["image_feat"], d["text_feat"], scheme=scheme, chunks=chunks,
                            replicate_feats=repl).to(dev).train()
    if scheme == "halo":
        model.halo = md.HaloPlan(need[0], ish, None, bk, dev)
    step = md.ShardedHotPathStep(model, graphs, 48, I, modal_empty=(modal.replace("_drop", "") == "empty_shortcut"),
                                 optimizer=False)
    step.set_batch(torch.stack([users, pos, neg]).to(dev))
    if drop:
        rows = (lambda k: md._pad_rows(k, ish.n_pad)) if repl else ish.slice_rows
        step.keep_masks = tuple(rows(k.to(torch.uint8)).to(dev) for k in T._global_masks(I))
    total = step.backward()
    torch.cuda.synchronize()
    assert model.last_fused
    g = {n: (p.grad.detach().cpu().clone() if p.grad is not None else None) for n, p in model.named_parameters()}
    assert float(g["txt_w"][:, k_txt:].abs().max()) == 0.0
    g["txt_w"] = g["txt_w"][:, :k_txt]
    rec = {"loss": float(total), "ush": (ush.lo, ush.hi, ush.n), "ish": (ish.lo, ish.hi, ish.n), "g": g,
           "chunks": model.n_chunks(2) if scheme in ("item-side", "halo") else 1}
    torch.save(_finish_peer(md, step, total, model, rec), os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def baby(rank, world, scheme, chunks, out_dir, md, dev, repl=False):
    """BASELINE configs[3]: the Amazon-Baby graph cut `world` ways (35598 users / 18357 items: uneven last blocks), one
    sharded step with injected dropout masks; the parent holds the oracle's loss and gradients."""
    import scipy.sparse as sp
    import numpy as np
    import _nccl_worker as W
    pb = W._baby_problem(torch.device("cpu"), ref=False)
    U, I = pb["U"], pb["I"]
    ush, ish = md.RowShard(U, world, rank), md.RowShard(I, world, rank)
    bk = md.HipBackend()
    ui_l = md.shard_graph(pb["ui"], ush, ish)
    iu_l = md.shard_graph_cols(pb["iu"], ish, ush) if scheme in ("item-side", "halo") else md.shard_graph(pb["iu"], ish, ush)
    need = None
    if scheme == "halo":
        need, ui_l, iu_l = md.halo_graphs(ui_l, iu_l)
    e_ui = bk.make_graph(sp.csr_matrix((ush.per, ish.n_pad), dtype=np.float32))
    e_iu = bk.make_graph(sp.csr_matrix((ish.per, ush.n_pad), dtype=np.float32))
    graphs = (bk.make_graph(ui_l), bk.make_graph(iu_l), e_ui, e_iu, e_ui, e_iu)
    model = md.ShardedMMSSL(bk, pb["cfg"], ush, ish, pb["state"], pb["img"].numpy(), pb["txt"].numpy(), scheme=scheme,
                            chunks=chunks, replicate_feats=repl).to(dev).train()
    if scheme == "halo":
        model.halo = md.HaloPlan(need, ish, None, bk, dev)
    step = md.ShardedHotPathStep(model, graphs, 1024, I, modal_empty=True, optimizer=False)
    rows = (lambda k: md._pad_rows(k, ish.n_pad)) if repl else ish.slice_rows
    step.keep_masks = tuple(rows(k.to(torch.uint8)).to(dev) for k in pb["km"])
    step.set_batch(pb["batch"].to(dev))
    total = step.backward()
    torch.cuda.synchronize()
    assert model.last_fused
    g = {n: (p.grad.detach().cpu().clone() if p.grad is not None else None) for n, p in model.named_parameters()}
    rec = {"loss": float(total), "ush": (ush.lo, ush.hi, ush.n), "ish": (ish.lo, ish.hi, ish.n), "g": g,
           "chunks": model.n_chunks(2) if scheme in ("item-side", "halo") else 1,
           "halo_fraction": (model.halo.bytes_fraction if scheme == "halo" else None)}
    torch.save(_finish_peer(md, step, total, model, rec), os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def synth_full(rank, world, scheme, chunks, out_dir, md, dev):
    """BASELINE configs[4] at its real size over `world` = 8 ranks: this rank generates ONLY its own 250 000 users' edges
    (dist.build_sharded_graph: the items' degrees are summed over the group), takes its rows of the seeded inputs
    (synth.stress_inputs) and runs one sharded step with the injected dropout masks. Saved: the job's loss, the small
    parameters' gradients, and this rank's table-gradient rows at the golden file's sampled row ids."""
    import types
    import numpy as np
    import scipy.sparse as sp
    import mmssl_oracle as O
    import helpers as H
    from mmssl_amd import synth
    a = types.SimpleNamespace(workload="synth", d=128, gcn_layers=3, batch=1024, scheme=scheme, chunks=chunks)
    ui_l, iu_l, ush, ish, U, I, E_global, _, _ = md.build_sharded_graph(a, rank, world, dev, "weak", scheme)
    assert (U, I) == (2_000_000, 1_000_000)
    bk = md.HipBackend()
    graphs = [bk.make_graph(ui_l), bk.make_graph(iu_l)]
    del ui_l, iu_l
    e_ui = bk.make_graph(sp.csr_matrix((ush.per, ish.n_pad), dtype=np.float32))
    e_iu = bk.make_graph(sp.csr_matrix((ish.per, ush.n_pad), dtype=np.float32))
    pb = synth.stress_inputs(U, I)
    cfg = O.Cfg(embed_size=128, n_ui_layers=3, drop_rate=0.2, batch_size=1024)
    repl = md.choose_replicate_feats(ish.n_pad, [pb["img"].shape[1], pb["txt"].shape[1]], 128, world)     # narrow features: True
    model = md.ShardedMMSSL(bk, cfg, ush, ish, pb["state"], pb["img"].numpy(), pb["txt"].numpy(), scheme=scheme,
                            chunks=chunks, replicate_feats=repl).to(dev).train()
    step = md.ShardedHotPathStep(model, tuple(graphs) + (e_ui, e_iu, e_ui, e_iu), 1024, I, modal_empty=True, optimizer=False)
    step.keep_masks = tuple((k if repl else ish.slice_rows(k)).to(dev) for k in pb["keep"])
    step.set_batch(pb["batch"].to(dev))
    del pb
    total = step.backward()
    torch.cuda.synchronize()
    assert model.last_fused
    z = H.load("synth_full_n1.npz")
    g = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if n not in ("E_u", "E_i") and p.grad is not None}
    for n, rows_k, sh in (("E_u", "rows_u", ush), ("E_i", "rows_i", ish)):
        rows = z[rows_k]
        loc = rows[(rows >= sh.lo) & (rows < sh.hi)] - sh.lo
        g[n] = dict(model.named_parameters())[n].grad[torch.from_numpy(loc).to(dev)].cpu()
    torch.save({"loss": float(total), "ush": (ush.lo, ush.hi, ush.n), "ish": (ish.lo, ish.hi, ish.n), "g": g,
                "edges_global": int(E_global), "chunks": model.n_chunks(2), "replicate_feats": bool(repl)},
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
