"""HBM-resident SpMM (configs[4]'s rank operand: A_ui[U_r, :] 250 000 x 1 000 000, 12.5 M edges, d = 128; the gathered table
is 512 MB = 2x the Infinity Cache): does a COLUMN-BANDED launch order pay? The columns are cut into nb contiguous bands,
A[:, band] gets its own plan, and Y = sum_k A[:, band_k] . X[band_k] is nb launches (the first plain, the others with the
AXPY epilogue accumulating into Y): every launch gathers from <= 512 / nb MB of the table, which then fits the Infinity
Cache, at the price of 2 (nb - 1) more passes over Y. Also: gather throughput against table size (one launch over the
first n columns' edges), to see what a cache-resident table is worth.     python tools/spmm_band_probe.py
Prints one JSON object (accept banding only if the forward drops >= 15 %)."""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def time_us(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    from mmssl_amd import ops, synth
    from mmssl_amd.graph import GraphPlan
    d, U_r, I = 128, 250_000, 1_000_000
    raw = synth.interaction_matrix(U_r, I, 12_500_000, seed=11)
    A = synth.normalised_rows(raw).tocsr()
    X = torch.randn(I, d, device="cuda")
    P = GraphPlan(A, xcd_bands=-1)
    out = {"shape": [U_r, I, int(A.nnz), d]}
    with torch.no_grad():
        Y0 = ops.spmm(P, X)
        out["whole_us"] = round(time_us(lambda: ops.spmm(P, X)), 1)
        Ac = A.tocsc()
        for nb in (2, 4, 8):
            w = I // nb
            plans = [GraphPlan(Ac[:, k * w:(k + 1) * w].tocsr(), xcd_bands=-1) for k in range(nb)]
            Xs = [X[k * w:(k + 1) * w] for k in range(nb)]
            Y = torch.empty(U_r, d, device="cuda")

            def banded():
                ops._spmm_raw(plans[0], False, Xs[0], ops.EPI_NONE, out=Y)
                for k in range(1, nb):
                    ops._spmm_raw(plans[k], False, Xs[k], ops.EPI_AXPY, Y, 1.0, out=Y)
            banded()
            err = float((Y - Y0).abs().max() / Y0.abs().max())
            out["bands_%d" % nb] = {"us": round(time_us(banded), 1), "max_rel_diff_vs_whole": float("%.3g" % err),
                                    "table_MB_per_band": round(w * d * 4e-6, 1)}
            del plans
        # gather throughput against the size of the gathered table: the edges whose column falls into the first n columns
        sizes = {}
        for n in (31_250, 62_500, 125_000, 250_000, 500_000, 1_000_000):
            sub = Ac[:, :n].tocsr()
            Pn = GraphPlan(sub, xcd_bands=-1)
            us = time_us(lambda: ops.spmm(Pn, X[:n]))
            by = synth.spmm_bytes(sub, d)
            sizes["%d_cols_%dMB" % (n, n * d * 4 // 1_000_000)] = {"edges": int(sub.nnz), "us": round(us, 1),
                                                                   "algorithmic_GBps": round(by / us * 1e-3, 1),
                                                                   "edges_per_us": round(sub.nnz / us, 1)}
            del Pn
        out["gather_vs_table_size"] = sizes
    print(json.dumps(out))


if __name__ == "__main__":
    main()
