"""Is ShardedHotPathStep.backward() idempotent on the HIP backend? (repeated calls on unchanged inputs must give bit-identical
losses and gradients.)  python tests/_repeat_probe.py WORLD SCHEME CHUNKS MODAL (a probe, not collected by pytest; it lives under tests/ because it checks against oracle/)   - ranks share GPU 0 over gloo."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, scheme, chunks, modal):
    import test_dist_cpu as T
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import mmssl_oracle as O
    from mmssl_amd import dist as md
    dev = torch.device("cuda", 0)
    fx, d, raw, U, I, state, users, pos, neg, img_raw, txt_raw = T._global_problem(modal)
    ush, ish = md.RowShard(U, world, rank), md.RowShard(I, world, rank)
    bk = md.HipBackend()
    cfg = O.Cfg(drop_rate=0.2, batch_size=48, n_ui_layers=2)
    d, state, k_txt = T._pad_text_to_slices(d, state)

    def row_pair(m):
        ui, iu = O.csr_norm(m, True), O.csr_norm(m.T, True)
        return bk.make_graph(md.shard_graph(ui, ush, ish)), bk.make_graph(md.shard_graph(iu, ish, ush))
    graphs = T._local_pair(md, bk, O, raw, ush, ish, scheme, []) + row_pair(img_raw) + row_pair(txt_raw)
    model = md.ShardedMMSSL(bk, cfg, ush, ish, state, d["image_feat"], d["text_feat"], scheme=scheme, chunks=chunks).to(dev).train()
    step = md.ShardedHotPathStep(model, graphs, 48, I, modal_empty=(modal == "empty_shortcut"), optimizer=False)
    step.set_batch(torch.stack([users, pos, neg]).to(dev))
    step.keep_masks = tuple(ish.slice_rows(k.to(torch.uint8)).to(dev) for k in T._global_masks(I))
    g0 = None
    for k in range(4):
        tot = step.backward()
        torch.cuda.synchronize()
        g = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        if g0 is None:
            g0, t0 = g, float(tot)
        else:
            print("rank", rank, "call", k, "loss", float(tot), "first", t0,
                  {n: float((g[n] - g0[n]).abs().max()) for n in g}, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import test_dist_cpu as T
    world, scheme, chunks, modal = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), sys.argv[4]
    mp.spawn(worker, args=(world, T._free_port(), scheme, chunks, modal), nprocs=world, join=True)
