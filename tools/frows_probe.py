"""The f-row launches (u_sim forward, the evaluation block) on the Baby shape, for rocprofv3 --kernel-trace --stats:
    cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d out -o f -- python tools/frows_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops, synth  # noqa: E402
from mmssl_amd.graph import GraphPlan  # noqa: E402

U, I, E, _, _ = synth.SHAPES["baby"]
d, B = 64, 1024
raw = synth.interaction_matrix(U, I, E, seed=1)
ui, _ = synth.normalised_pair(raw)
plan = GraphPlan(ui)
dev = torch.device("cuda")
g = torch.Generator().manual_seed(11)
ua, ia = torch.randn(U, d, generator=g).to(dev), torch.randn(I, d, generator=g).to(dev)
users = torch.randperm(U, generator=g)[:B].to(dev)
rp = torch.from_numpy(raw.indptr.astype(np.int32)).to(dev)
cols = torch.from_numpy(raw.indices.astype(np.int32)).to(dev)
acc = torch.zeros((4, 8), dtype=torch.float64, device=dev)
ws = None
with torch.no_grad():
    for _ in range(int(os.environ.get("ITERS", "20"))):
        ops.usim(users, ua, ia, plan)
        rate, _ = ops.sim_rows(ua, ia, qidx=users, mask=(rp, cols), mask_value=float("-inf"))
        order = ops.topk_rows(rate, 50)
        ws = ops.eval_accumulate(rp, cols, users, order, [10, 20, 50], acc, ws)
torch.cuda.synchronize()


def _us(fn, iters=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / iters, 1)


if os.environ.get("MODE") == "time":       # each launch group on its own, HIP events around 100 back-to-back calls
    import json
    with torch.no_grad():
        rate, _ = ops.sim_rows(ua, ia, qidx=users, mask=(rp, cols), mask_value=float("-inf"))
        order = ops.topk_rows(rate, 50)
        rec = {"lib": os.path.basename(os.environ.get("MMSSL_LIB", "libmmssl_hip.so")),
               "usim_forward_us": _us(lambda: ops.usim(users, ua, ia, plan)),
               "scores_minus_inf_mask_us": _us(lambda: ops.sim_rows(ua, ia, qidx=users, mask=(rp, cols), mask_value=float("-inf"))),
               "scores_no_mask_us": _us(lambda: ops.sim_rows(ua, ia, qidx=users)),
               "topk50_us": _us(lambda: ops.topk_rows(rate, 50)),
               "metrics_us": _us(lambda: ops.eval_accumulate(rp, cols, users, order, [10, 20, 50], acc, ws))}
    print(json.dumps(rec))

if os.environ.get("MODE") == "tiles":      # only the bitmap-free tile kernel, for rocprofv3 --stats per build
    with torch.no_grad():
        for _ in range(30):
            ops.sim_rows(ua, ia, qidx=users)
            ops.sim_rows(ua, ia, qidx=users, pitch_mult=32)
    torch.cuda.synchronize()
