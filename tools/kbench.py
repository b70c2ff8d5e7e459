"""Kernel micro-benchmarks on one MI355X: per-kernel time (HIP events on the launch stream),
algorithmic GB/s vs the 8 TB/s HBM roofline, TFLOP/s vs the 157.3 TF fp32 MFMA peak.
    python tools/kbench.py [--shape baby] [--d 64]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import graph, ops, synth  # noqa: E402


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters   # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="baby")
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--edges", type=int, default=0)
    a = ap.parse_args()
    U, I, E, dv, dt = synth.SHAPES[a.shape]
    if a.edges:
        E = a.edges
    d = a.d
    t0 = time.time()
    raw = synth.interaction_matrix(U, I, E)
    ui, iu = synth.normalised_pair(raw)
    print("graph %s: U=%d I=%d E=%d (%.1fs) max_deg ui=%d iu=%d" % (
        a.shape, U, I, raw.nnz, time.time() - t0, np.diff(ui.indptr).max(), np.diff(iu.indptr).max()))
    t0 = time.time()
    P_ui, P_iu = graph.GraphPlan(ui), graph.GraphPlan(iu)
    print("plans built in %.3fs" % (time.time() - t0), P_ui.info(), P_iu.info())
    dev = "cuda"
    Xi = torch.randn(I, d, device=dev)
    Xu = torch.randn(U, d, device=dev)
    out = {}

    def report(name, us, nbytes=None, flops=None):
        rec = {"us": round(us, 2)}
        if nbytes:
            rec["GBps"] = round(nbytes / us * 1e-3, 1)
            rec["frac_hbm"] = round(nbytes / us * 1e-3 / 8000.0, 3)
        if flops:
            rec["TFLOPs"] = round(flops / us * 1e-6, 2)
            rec["frac_mfma"] = round(flops / us * 1e-6 / 157.3, 3)
        out[name] = rec
        print("%-28s %s" % (name, json.dumps(rec)))

    with torch.no_grad():
        b_ui, b_iu = synth.spmm_bytes(ui, d), synth.spmm_bytes(iu, d)
        report("spmm A_ui.Xi", timeit(lambda: ops.spmm(P_ui, Xi)), b_ui)
        report("spmm A_iu.Xu", timeit(lambda: ops.spmm(P_iu, Xu)), b_iu)
        report("spmm A_ui.Xi softmax", timeit(lambda: ops.spmm(P_ui, Xi, ops.EPI_SOFTMAX)), b_ui)
        report("spmm A_ui^T.Xu (bwd)", timeit(lambda: ops.spmm(P_ui, Xu, transpose=True)), synth.spmm_bytes(ui.T.tocsr(), d))

        def gcn3():
            u, i = Xu, Xi
            for l in range(3):
                epi = ops.EPI_SOFTMAX if l == 2 else ops.EPI_NONE
                u = ops.spmm(P_ui, i, epi)
                i = ops.spmm(P_iu, u, epi)
            return u, i
        us = timeit(gcn3, iters=100)
        report("gcn 3-layer fwd (6 spmm)", us, 3 * (b_ui + b_iu))
        out["gcn 3-layer fwd (6 spmm)"]["G_edge_layers_per_s"] = round(6 * raw.nnz / us * 1e-3, 2)
        print("   -> %.2f G edge.layers/s (eager launches)" % (6 * raw.nnz / us * 1e-3))
        # same chain under a hipGraph (no host launch gaps)
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            gcn3()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                gcn3()
        us = timeit(g.replay, iters=100)
        report("gcn 3-layer fwd hipGraph", us, 3 * (b_ui + b_iu))
        print("   -> %.2f G edge.layers/s (hipGraph)" % (6 * raw.nnz / us * 1e-3))
        report("l2norm rows [U,d]", timeit(lambda: ops.l2norm_rows(Xu)), 2 * U * d * 4)

        if dv:
            for nm, K in (("img", dv), ("txt", dt)):
                F_ = torch.randn(I, K, device=dev)
                W = torch.randn(d, K, device=dev) * 0.02
                b = torch.zeros(d, device=dev)
                keep = (torch.rand(I, d, device=dev) >= 0.2).to(torch.uint8)
                fl = 2.0 * I * K * d
                by = 4.0 * (I * K + K * d + I * d)
                report("linear %s fwd K=%d" % (nm, K), timeit(lambda: ops.linear(F_, W, b, keep, 1.25), iters=50), by, fl)
                gY = torch.randn(I, d, device=dev)
                from mmssl_amd import _lib
                gW = torch.empty_like(W); gb = torch.empty(d, device=dev)
                nb = _lib.lib().mmssl_linear_wgrad_workspace_bytes(I, K, d)
                ws = torch.empty(nb // 4, device=dev)

                def wg():
                    _lib.lib().mmssl_linear_wgrad_f32(gY.data_ptr(), None, 1.0, F_.data_ptr(), I, K, d, gW.data_ptr(), gb.data_ptr(),
                                                      ws.data_ptr(), nb, _lib.stream_ptr())
                report("linear %s wgrad K=%d" % (nm, K), timeit(wg, iters=50), by, fl)
                del F_
    B = 1024
    z1 = torch.randn(B, d, device=dev, requires_grad=True)
    z2 = torch.randn(B, d, device=dev, requires_grad=True)
    report("infonce fwd B=1024", timeit(lambda: ops.infonce(z1.detach(), z2.detach(), 0.5)), None, 2 * 2 * B * B * d)

    def nce_fb():
        l = ops.infonce(z1, z2, 0.5)
        l.backward()
        z1.grad = None; z2.grad = None
    report("infonce fwd+bwd B=1024", timeit(nce_fb, iters=100))
    users = torch.randperm(U, device=dev)[:B]
    pos = torch.randint(0, I, (B,), device=dev)
    neg = torch.randint(0, I, (B,), device=dev)
    with torch.no_grad():
        report("bpr fwd (fused gather)", timeit(lambda: ops.bpr_gather(Xu, Xi, users, pos, neg, 1e-5, B)))
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/kbench_%s_d%d.json" % (a.shape, d), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
