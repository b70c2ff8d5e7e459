"""Child rank of tests/test_peer_gpu.py: the peer-exchange primitives (mmssl_amd/peer.py over csrc/peer.hip) with `world`
processes sharing GPU 0 - IPC-mapped windows, epoch flags, push / wait / pull-sum - against values every rank can compute
for itself. Eager steps first (windows are created on first use), then the same sequence captured in a hipGraph and
replayed with new inputs.

    python tests/_peer_worker.py RANK WORLD PORT [timeout | setupfail] OUT_DIR"""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def shard(rank, step, per, w):
    g = torch.Generator().manual_seed(1000 * step + rank)
    return torch.randn(per, w, generator=g)


def partial(rank, step, rows, w):
    g = torch.Generator().manual_seed(77000 + 1000 * step + rank)
    return torch.randn(rows, w, generator=g)


def main():
    rank, world, port, out_dir = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[-1]
    mode = sys.argv[4] if len(sys.argv) > 5 else "exchange"
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mmssl_amd import peer
    dev = torch.device("cuda", 0)
    if mode == "timeout":
        # rank 0 waits on a channel rank 1 never signals: the wait gives up after 300 ms (the device is not hung), the
        # error word says which peer was missing and check() raises
        from mmssl_amd import _lib
        t = peer.PeerTransport(dist.group.WORLD, dev, timeout_ms=300)
        rec = {"timed_out": False}
        if rank == 0:
            t.signal(5)
            t.wait(5)
            torch.cuda.synchronize()
            try:
                t.check()
            except _lib.MmsslError as e:
                rec = {"timed_out": True, "msg": str(e)}
        torch.save(rec, os.path.join(out_dir, "r%d.pt" % rank))
        dist.barrier()
        t.close()
        dist.destroy_process_group()
        return
    if mode == "setupfail":
        # the window export fails on the LAST rank only: the failure travels through the handle exchange, every rank raises
        # the same error at the same point (no rank is left alone in a host-side collective) and the job goes on - over
        # torch.distributed, as bench.choose_transport does
        from mmssl_amd import _lib
        pc = peer.PeerComm(dist.group.WORLD, dev, timeout_ms=2000)
        if rank == world - 1:
            real = pc.t._L.mmssl_peer_window_create

            def broken(*a):
                return -1
            pc.t._L = type("L", (), {"__getattr__": lambda self_, k: broken if k == "mmssl_peer_window_create"
                                     else getattr(_lib.lib(), k)})()
            del real
        rec = {"raised": False}
        try:
            pc.begin_step()
            pc.gather(torch.ones(4, 64, device=dev))
        except _lib.MmsslError as e:
            rec = {"raised": True, "msg": str(e)}
        t = torch.ones(1) * (rank + 1)
        dist.all_reduce(t)                              # the ranks are still in step with each other
        rec["sum"] = float(t)
        torch.save(rec, os.path.join(out_dir, "r%d.pt" % rank))
        dist.barrier()
        pc.t._L = _lib.lib()
        pc.close()
        dist.destroy_process_group()
        return
    pc = peer.PeerComm(dist.group.WORLD, dev, timeout_ms=60000)
    per, w, n_small = 37, 64, 1001                      # an odd row count, a small buffer that is not a multiple of 4
    rows = world * per
    wide = torch.zeros(per, 2 * w, device=dev)          # a row-pitched shard: the right half of a wider table
    x_in = torch.zeros(per, w, device=dev)
    p_in = torch.zeros(rows, w, device=dev)
    s_in = torch.zeros(n_small, device=dev)

    def sequence():
        pc.begin_step()
        full = pc.gather(x_in)                           # [rows, w]: the window itself
        full2 = pc.gather(wide[:, w:])
        red = pc.reduce(p_in, per)                       # copy-in form
        pw = pc.partial(rows, w)
        pw.copy_(p_in).mul_(2.0)                         # "the SpMM wrote its output here"
        red2 = pc.reduce(pw, per)
        small = s_in.clone()
        pc.all_reduce_(small)
        return full, full2, red, red2, small

    def load(step):
        x_in.copy_(shard(rank, step, per, w))
        wide[:, w:].copy_(shard(rank, step, per, w) * 3.0)
        p_in.copy_(partial(rank, step, rows, w))
        s_in.copy_(torch.arange(n_small, dtype=torch.float32) * (rank + 1) + step)

    def check(step, outs):
        full, full2, red, red2, small = [o.cpu() for o in outs]
        want_full = torch.cat([shard(q, step, per, w) for q in range(world)], 0)
        assert torch.equal(full, want_full), ("gather", step)
        assert torch.equal(full2, want_full * 3.0), ("pitched gather", step)
        acc = partial(0, step, rows, w)[rank * per:(rank + 1) * per].clone()
        for q in range(1, world):                        # rank order 0, 1, ...: the kernel's order, so the same bits
            acc += partial(q, step, rows, w)[rank * per:(rank + 1) * per]
        assert torch.equal(red, acc), ("reduce", step, float((red - acc).abs().max()))
        acc2 = 2.0 * partial(0, step, rows, w)[rank * per:(rank + 1) * per]
        for q in range(1, world):
            acc2 += 2.0 * partial(q, step, rows, w)[rank * per:(rank + 1) * per]
        assert torch.equal(red2, acc2), ("reduce from the window", step)
        ws = torch.arange(n_small, dtype=torch.float32) * 1 + step
        for q in range(1, world):
            ws = ws + (torch.arange(n_small, dtype=torch.float32) * (q + 1) + step)
        assert torch.equal(small, ws), ("all_reduce", step)

    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for step in range(4):
            load(step)
            outs = sequence()
            s.synchronize()
            check(step, outs)
        sites, wins = pc.stats()["call_sites"], pc.stats()["windows"]
        # the same sequence as ONE hipGraph: pointers and channels are fixed, the epochs live in device memory
        g = torch.cuda.CUDAGraph()
        load(4)
        with torch.cuda.graph(g, stream=s):
            outs = sequence()
        for step in range(5, 9):
            load(step)
            g.replay()
            s.synchronize()
            check(step, outs)
    pc.check()
    st = pc.stats()
    assert st["call_sites"] == sites and st["windows"] == wins            # nothing was created after the first step
    torch.save({"stats": st}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    pc.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
