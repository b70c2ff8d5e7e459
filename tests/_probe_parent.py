"""torchrun target for test_dist_cpu.py::test_spawn_rank_probe (gloo): every rank spawns a child via
mmssl_amd.dist.spawn_rank_probe; the children rendezvous among themselves on MASTER_PORT+1."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd.dist import spawn_rank_probe  # noqa: E402


def child():
    dist.init_process_group("gloo")
    t = torch.tensor([dist.get_rank() + 1.0])
    dist.all_reduce(t)
    w = dist.get_world_size()
    ok = float(t.item()) == w * (w + 1) / 2
    dist.destroy_process_group()
    # rank 1's child fails on purpose in "fail" mode: every parent must then see False after the MIN
    if sys.argv[2] == "fail" and os.environ["RANK"] == "1":
        ok = False
    sys.exit(0 if ok else 3)


def parent():
    dist.init_process_group("gloo")
    mode = sys.argv[2]
    mine = spawn_rank_probe([sys.executable, os.path.abspath(__file__), "child", mode], timeout=120)
    flag = torch.tensor([1 if mine else 0], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    agreed = bool(flag.item())
    expect = mode == "ok"
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if agreed == expect else 5)


if __name__ == "__main__":
    child() if sys.argv[1] == "child" else parent()
