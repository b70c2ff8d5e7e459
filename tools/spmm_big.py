"""HBM-resident SpMM stress: tables far larger than L2 (and ~Infinity-Cache sized), d=128 like the
BASELINE 'synth' shape (one rank's share of it).  python tools/spmm_big.py [U I E d]"""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import graph, ops, synth
U, I, E, d = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (500_000, 250_000, 25_000_000, 128)))
t0 = time.time(); raw = synth.interaction_matrix(U, I, E); ui, iu = synth.normalised_pair(raw)
print("graph U=%d I=%d E=%d built in %.1fs; max deg ui=%d iu=%d" % (U, I, raw.nnz, time.time() - t0, np.diff(ui.indptr).max(), np.diff(iu.indptr).max()))
t0 = time.time(); P_ui, P_iu = graph.GraphPlan(ui), graph.GraphPlan(iu); print("plans %.1fs" % (time.time() - t0), P_iu.info())
Xi, Xu = torch.randn(I, d, device="cuda"), torch.randn(U, d, device="cuda")
out = {}
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
with torch.no_grad():
    for nm, P, X, m in (("A_ui.Xi", P_ui, Xi, ui), ("A_iu.Xu", P_iu, Xu, iu), ("A_ui^T.Xu", P_ui, Xu, ui.T.tocsr()), ("A_iu^T.Xi", P_iu, Xi, iu.T.tocsr())):
        tr = "^T" in nm
        us = t(lambda: ops.spmm(P, X, transpose=tr))
        by = synth.spmm_bytes(m, d)
        out[nm] = {"us": round(us, 1), "alg_GB": round(by / 1e9, 2), "GBps": round(by / us * 1e-3, 1), "frac_hbm": round(by / us * 1e-3 / 8000, 3)}
        print(nm, out[nm])
    # sampled correctness vs scipy
    Y = ops.spmm(P_ui, Xi).cpu().numpy(); rows = np.random.default_rng(0).choice(U, 500, replace=False)
    ref = (ui[rows] @ Xi.cpu().numpy())
    print("sample max rel err", float(np.abs(Y[rows] - ref).max() / np.abs(ref).max()))
os.makedirs("gpurun_out", exist_ok=True); json.dump(out, open("gpurun_out/spmm_big.json", "w"), indent=1)
