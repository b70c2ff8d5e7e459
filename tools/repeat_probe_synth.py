"""Idempotence of ShardedHotPathStep.backward() on the configs[4] rank share (d = 128, per-modality projection, column chunks,
replicated features): repeated calls on unchanged inputs.   python tools/repeat_probe_synth.py WORLD CHUNKS REPL(on|off)"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, chunks, repl):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    if os.environ.get("PROBE_BACKEND") == "nccl":       # one rank, real RCCL launches of the self-collectives (stream-ordered)
        os.environ["MMSSL_DIST_FORCE_COLLECTIVES"] = "1"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmssl_amd import dist as md
    dev = torch.device("cuda", 0)
    a = types.SimpleNamespace(workload="synth", d=128, gcn_layers=3, batch=1024, scheme="item-side", chunks=chunks,
                              replicate_feats=repl)
    step0, mats, plans, stats = md.build_bench_step(a, rank, world, dev, "weak")
    model = step0.model
    step = md.ShardedHotPathStep(model, step0.graphs, 1024, stats["n_items"], modal_empty=True, optimizer=False)
    g = torch.Generator().manual_seed(5)
    rows = model.image_feats.shape[0]
    step.keep_masks = tuple((torch.rand(rows, 128, generator=g) >= 0.2).to(torch.uint8).to(dev) for _ in range(2))
    b = torch.stack([torch.randperm(stats["n_users"], generator=g)[:1024], torch.randint(0, stats["n_items"], (1024,), generator=g),
                     torch.randint(0, stats["n_items"], (1024,), generator=g)])
    step.set_batch(b.to(dev))
    if os.environ.get("PRE_FWD"):        # two forwards without autograd first (an evaluation before training, say)
        with torch.no_grad():
            for _ in range(2):
                model(step.graphs, keep_masks=torch.stack(step.keep_masks), modal_empty=True)
        torch.cuda.synchronize()
    snap = {n: p.detach().clone() for n, p in model.named_parameters()}
    g0 = None
    for k in range(4):
        tot = step.backward()
        torch.cuda.synchronize()
        gr = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        moved = {n: float((p.detach() - snap[n]).abs().max()) for n, p in model.named_parameters()}
        if g0 is None:
            g0, t0 = gr, float(tot)
        elif rank == 0:
            print("call", k, "loss", float(tot), "first", t0, "grad diff", {n: float("%.3g" % float((gr[n] - g0[n]).abs().max())) for n in gr},
                  "params moved", {n: v for n, v in moved.items() if v > 0}, flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import test_dist_cpu as T
    world, chunks, repl = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    mp.spawn(worker, args=(world, T._free_port(), chunks, repl), nprocs=world, join=True)
