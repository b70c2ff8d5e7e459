"""FETCH_SIZE / WRITE_SIZE calibration for ROW GATHERS (the SpMM's access pattern) on gfx950.

The microarchitecture guide calibrates rocprofv3's FETCH_SIZE only for wide streaming reads (it reports 1/2 of the
bytes there) and says other access widths are uncalibrated. Here the SpMM kernel itself gathers a KNOWN number of
bytes: A is a permutation matrix (one edge per row, every column referenced exactly once), X is >= 1 GB (4x the
256 MB Infinity Cache, so nothing is served on-die), so one launch reads exactly rows * 4d bytes of X in 4d-byte
pieces at random addresses + 8 B per edge + the work list, and writes rows * 4d bytes.

    MODE=run python tools/fetch_calib.py            # prints the known byte counts (JSON)
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o f -- python tools/fetch_calib.py
    rocprofv3 --pmc WRITE_SIZE ...                  # separate pass
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import graph, ops  # noqa: E402

known = {}
for d, rows in ((64, 4 << 20), (128, 2 << 20)):
    rng = np.random.default_rng(d)
    perm = rng.permutation(rows).astype(np.int32)
    A = sp.csr_matrix((np.ones(rows, np.float32), perm, np.arange(rows + 1, dtype=np.int32)), shape=(rows, rows))
    P = graph.GraphPlan(A)
    X = torch.randn(rows, d, device="cuda")
    info = P.info()
    with torch.no_grad():
        for _ in range(4):
            Y = ops.spmm(P, X)
        torch.cuda.synchronize()
    known["spmm_kernel<%d" % (d // 4)] = {
        "d": d, "rows": rows, "gathered_X_bytes": rows * 4 * d, "edge_bytes": rows * 8,
        "item_bytes": 16 * (info["group_items"] + info["wave_items"]), "written_Y_bytes": rows * 4 * d,
        "read_total": rows * 4 * d + rows * 8 + 16 * (info["group_items"] + info["wave_items"])}
    del X, Y, P
print(json.dumps(known))
