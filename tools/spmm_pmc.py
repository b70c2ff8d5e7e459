"""SpMM-only driver for rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o f -- python tools/spmm_pmc.py
Launches the four SpMM flavours of the hot-path step (A_ui, A_iu, both transposes), 10 rounds."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import graph, ops, synth  # noqa: E402

U, I, E, _, _ = synth.SHAPES[os.environ.get("SHAPE", "baby")]
d = int(os.environ.get("D", "64"))
raw = synth.interaction_matrix(U, I, E)
ui, iu = synth.normalised_pair(raw)
P = (graph.GraphPlan(ui), graph.GraphPlan(iu))
Xi, Xu = torch.randn(I, d, device="cuda"), torch.randn(U, d, device="cuda")
with torch.no_grad():
    for _ in range(10):
        ops.spmm(P[0], Xi)
        ops.spmm(P[1], Xu)
        ops.spmm(P[0], Xu, transpose=True)
        ops.spmm(P[1], Xi, transpose=True)
torch.cuda.synchronize()
