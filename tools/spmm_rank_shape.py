"""One rank's share of BASELINE configs[4] (2M users x 1M items, 100M edges, d=128, 8-way row shard) on ONE GPU:
    A_ui_r : 250 000 local user rows x 1 000 000 global item columns, 12.5 M edges, X = gathered item table [1M, 128] (512 MB)
    A_iu_r : 125 000 local item rows x 2 000 000 global user columns, 12.5 M edges, X = gathered user table [2M, 128] (1 GB)
plus both transposes (the backward's partial A_r^T . gY, full-height output). Tables are 2-4x the 256 MiB Infinity
Cache, so this is the HBM-resident gather.  MODE=time (default): HIP-event timing -> gpurun_out/spmm_rank_shape.json;
MODE=pmc: 3 launches of each flavour for a rocprofv3 --pmc pass."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import graph, ops, synth  # noqa: E402

world = int(os.environ.get("WORLD", "8"))
d = int(os.environ.get("D", "128"))
U, I, E = 2_000_000, 1_000_000, 100_000_000
mode = os.environ.get("MODE", "time")
t0 = time.time()
# local user rows of A_ui: a [U/world, I] interaction block with E/world edges
raw_u = synth.interaction_matrix(U // world, I, E // world, seed=11)
ui_r = synth.normalised_pair(raw_u)[0]
# local item rows of A_iu: generate the [U, I/world] block and transpose it
raw_i = synth.interaction_matrix(U, I // world, E // world, seed=12)
iu_r = synth.normalised_pair(raw_i)[1]
print("shards built in %.1fs: A_ui_r %s nnz %d max_deg %d | A_iu_r %s nnz %d max_deg %d" % (
    time.time() - t0, ui_r.shape, ui_r.nnz, np.diff(ui_r.indptr).max(), iu_r.shape, iu_r.nnz, np.diff(iu_r.indptr).max()),
    flush=True)
t0 = time.time()
P_ui, P_iu = graph.GraphPlan(ui_r), graph.GraphPlan(iu_r)
print("plans %.1fs" % (time.time() - t0), P_ui.info(), P_iu.info(), flush=True)
Xi = torch.randn(I, d, device="cuda")
Xu = torch.randn(U, d, device="cuda")
gYu = torch.randn(U // world, d, device="cuda")
gYi = torch.randn(I // world, d, device="cuda")
cases = (("A_ui_r.Xi_full", P_ui, Xi, False, ui_r), ("A_iu_r.Xu_full", P_iu, Xu, False, iu_r),
         ("A_ui_r^T.gYu", P_ui, gYu, True, ui_r.T.tocsr()), ("A_iu_r^T.gYi", P_iu, gYi, True, iu_r.T.tocsr()))
out = {"d": d, "world": world}
with torch.no_grad():
    for nm, P, X, tr, m in cases:
        by = synth.spmm_bytes(m, d)
        once = m.nnz * 8 + (m.shape[0] + 1) * 4 + m.shape[0] * 4 * d + min(m.shape[1], m.nnz) * 4 * d
        if mode == "pmc":
            for _ in range(3):
                ops.spmm(P, X, transpose=tr)
            torch.cuda.synchronize()
            continue
        for _ in range(3):
            ops.spmm(P, X, transpose=tr)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.spmm(P, X, transpose=tr)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        out[nm] = {"us": round(us, 1), "rows": int(m.shape[0]), "cols": int(m.shape[1]), "nnz": int(m.nnz),
                   "algorithmic_GB": round(by / 1e9, 3), "algorithmic_GBps": round(by / us * 1e-3, 1),
                   "frac_hbm_algorithmic": round(by / us * 1e-3 / 8000, 3),
                   "once_through_GB": round(once / 1e9, 3), "once_through_GBps": round(once / us * 1e-3, 1),
                   "edge_layers_per_s": round(m.nnz / us * 1e6, 1)}
        print(nm, out[nm], flush=True)
    if mode != "pmc":
        Y = ops.spmm(P_ui, Xi).cpu().numpy()
        rows = np.random.default_rng(0).choice(ui_r.shape[0], 300, replace=False)
        ref = ui_r[rows] @ Xi.cpu().numpy()
        out["sample_max_rel_err"] = float(np.abs(Y[rows] - ref).max() / np.abs(ref).max())
        print("sample max rel err", out["sample_max_rel_err"])
        os.makedirs("gpurun_out", exist_ok=True)
        json.dump(out, open("gpurun_out/spmm_rank_shape.json", "w"), indent=1)
