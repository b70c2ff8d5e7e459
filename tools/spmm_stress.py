"""Is the SpMM's in-kernel combine of multi-block rows (write-through partial slots + arrival ticket, csrc/graph.hip) safe
when the GPU is shared by several processes? N processes each run the configs[4] rank operand's TRANSPOSED product (hub
items: rows of ~10^5 edges spanning hundreds of partial slots) in a loop and compare every launch bit for bit with their
first one, and the 64 longest rows with a float64 CPU product.   python tools/spmm_stress.py [N=8] [iters=30] [chunks=1]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, iters, width):
    import numpy as np
    import torch
    from mmssl_amd import ops, synth
    from mmssl_amd.graph import GraphPlan
    torch.cuda.set_device(0)
    raw = synth.interaction_matrix(250_000, 1_000_000, 12_500_000, seed=1000 + rank, item_seed=77)
    A = synth.normalised_rows(raw)
    P = GraphPlan(A, xcd_bands=-1)
    AT = A.T.tocsr()
    deg = np.diff(AT.indptr)
    rows = np.argsort(deg)[-64:]
    g = torch.Generator().manual_seed(rank)
    G = torch.randn(250_000, width, generator=g)
    Gd = G.cuda()
    ref = torch.from_numpy(np.asarray(AT[rows].astype(np.float64) @ G.double().numpy()))
    idx = torch.from_numpy(rows).cuda()
    first, bad_bits, worst = None, 0, 0.0
    streams = [torch.cuda.Stream() for _ in range(4)]
    for it in range(iters):
        outs = []
        for k, s in enumerate(streams):          # four concurrent launches per process, each with its own workspace lane
            with torch.cuda.stream(s):
                outs.append(ops._spmm_raw(P.twin(2 + k), True, Gd, ops.EPI_NONE))
        torch.cuda.synchronize()
        for Y in outs:
            if first is None:
                first = Y.clone()
            elif not torch.equal(Y, first):
                bad_bits += 1
            e = float((Y[idx].double().cpu() - ref).abs().max() / ref.abs().max())
            worst = max(worst, e)
    print(json.dumps({"rank": rank, "launches": iters * len(streams), "not_bit_equal_to_first": bad_bits,
                      "worst_rel_err_64_longest_rows": float("%.3g" % worst), "longest_row": int(deg.max())}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
        iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
        width = int(sys.argv[3]) if len(sys.argv) > 3 else 32
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(iters), str(width)]) for r in range(n)]
        sys.exit(max(p.wait() for p in ps))
