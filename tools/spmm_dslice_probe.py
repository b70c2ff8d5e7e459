"""VERDICT r5 item 8 (every XCD's L2 pulls its own copy of the gathered table: 43.4 MB of fabric traffic per Baby SpMM
launch against ~16 MB once-through): what would slicing the feature width buy? With the table cut into two 32-float
column halves, an XCD that only ever gathers ONE half works on 2.3 MB (fits its 4 MB L2) and the halves are duplicated
4 x instead of 8 x - but every edge is then processed twice. This probe measures the second part on the real kernels:
the same product as ONE d = 64 launch and as TWO d = 32 column-chunk launches (row-pitched operands,
mmssl_spmm_ld_f32), times from HIP events; run it under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` for the traffic.

    python tools/spmm_dslice_probe.py            MODE=pmc: 10 rounds of each form, no timing"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import graph, ops, synth  # noqa: E402

U, I, E, _, _ = synth.SHAPES["baby"]
raw = synth.interaction_matrix(U, I, E)
ui, iu = synth.normalised_pair(raw)
P = (graph.GraphPlan(ui), graph.GraphPlan(iu))
Xi, Xu = torch.randn(I, 64, device="cuda"), torch.randn(U, 64, device="cuda")
Yu, Yi = torch.empty(U, 64, device="cuda"), torch.empty(I, 64, device="cuda")


def whole():
    ops._spmm_raw(P[0], False, Xi, ops.EPI_NONE, out=Yu)
    ops._spmm_raw(P[1], False, Xu, ops.EPI_NONE, out=Yi)


def halves():
    for c in (0, 32):
        ops._spmm_raw(P[0], False, Xi[:, c:c + 32], ops.EPI_NONE, out=Yu[:, c:c + 32])
    for c in (0, 32):
        ops._spmm_raw(P[1], False, Xu[:, c:c + 32], ops.EPI_NONE, out=Yi[:, c:c + 32])


def us(fn, iters=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / iters / 2, 2)          # per product (two products per call)


with torch.no_grad():
    if os.environ.get("MODE") == "pmc":
        for _ in range(10):
            whole()
        for _ in range(10):
            halves()
        torch.cuda.synchronize()
    else:
        whole()
        a = Yu.clone()
        halves()
        same = bool(torch.equal(a, Yu))
        print(json.dumps({"one_d64_launch_us": us(whole), "two_d32_column_chunk_launches_us": us(halves),
                          "results_bit_equal": same}))
