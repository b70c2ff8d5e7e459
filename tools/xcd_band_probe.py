"""XCD-banded SpMM work list (GraphPlan(xcd_bands=...)) against the flat degree-sorted one, Baby-shaped graphs, d = 64 and
128: the uniform synthetic graph (no column locality by construction) and a community-structured one (8 communities, 10 %
global edges: synth.interaction_matrix_communities). HIP-event time per launch of the four flavours of a step (A_ui, A_iu,
both transposes) under hipGraph replay; MODE=pmc: 5 launches of each (plan, flavour) for a rocprofv3 --pmc pass
(kernel order: graph kind, then flat / banded, then flavour).

    python tools/xcd_band_probe.py > gpurun_out/xcd_band.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import graph, ops, synth  # noqa: E402

U, I, E, _, _ = synth.SHAPES["baby"]
mode = os.environ.get("MODE", "time")
out = {}
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
_com = synth.interaction_matrix_communities(U, I, E, seed=1)
_rng = np.random.default_rng(5)
_perm = sp.csr_matrix(_com[_rng.permutation(U)][:, _rng.permutation(I)])      # the same graph, randomly renumbered
for kind, raw in (("uniform", synth.interaction_matrix(U, I, E, seed=1)),
                  ("communities", _com), ("communities_renumbered", _perm)):
    ui, iu = synth.normalised_pair(raw)
    for d in ((64,) if mode == "pmc" else (64, 128)):
        Xi, Xu = torch.randn(I, d, device="cuda"), torch.randn(U, d, device="cuda")
        variants = [("flat", -1), ("banded", 2 if kind == "communities_renumbered" else 1)]
        if kind == "uniform" and mode != "pmc":
            variants.append(("coclustered", 2))          # forced co-clustering where the automatic rule says no (score 0.41)
        for name, bands in variants:
            P = (graph.GraphPlan(ui, xcd_bands=bands), graph.GraphPlan(iu, xcd_bands=bands))
            launches = [(P[0], False, Xi, ui), (P[1], False, Xu, iu), (P[0], True, Xu, ui.T.tocsr()), (P[1], True, Xi, iu.T.tocsr())]
            rec = {"score": [P[0].info()["band_score"] or P[0].info()["cluster_score"],
                             P[1].info()["band_score"] or P[1].info()["cluster_score"]]}
            with torch.no_grad():
                if mode == "pmc":
                    for (p, t, X, m) in launches:
                        for _ in range(5):
                            ops.spmm(p, X, transpose=t)
                    torch.cuda.synchronize()
                    continue
                for fl, (p, t, X, m) in zip(("A_ui", "A_iu", "A_ui^T", "A_iu^T"), launches):
                    for _ in range(5):
                        ops.spmm(p, X, transpose=t)
                    g = torch.cuda.CUDAGraph()
                    s = torch.cuda.Stream()
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        torch.cuda.synchronize()
                        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                            for _ in range(20):
                                ops.spmm(p, X, transpose=t)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    with torch.cuda.stream(s):
                        g.replay()
                        e0.record()
                        for _ in range(20):
                            g.replay()
                        e1.record()
                    torch.cuda.synchronize()
                    us = e0.elapsed_time(e1) * 1e3 / 400
                    by = synth.spmm_bytes(m, d)
                    rec[fl] = {"us": round(us, 2), "frac_hbm_algorithmic": round(by / us * 1e-3 / 8000, 3)}
            out["%s/d%d/%s" % (kind, d, name)] = rec
if mode != "pmc":
    print(json.dumps(out, indent=1))
