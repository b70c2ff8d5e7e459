"""The HBM-resident SpMM of bench.py's `spmm_hbm` record (configs[4] rank shape: A_ui[U_r, :] 250 000 x 1 000 000, 12.5 M
edges, d = 128; gathered table 512 MB) under rocprofv3 PMC passes:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o p -- python tools/spmm_hbm_pmc.py
    python tools/spmm_hbm_pmc.py summarise FETCH.csv WRITE.csv OUT.json

Run mode launches the forward product 3 x, then the transposed one 3 x (same graph seed as bench.py). Summarise writes
{forward, transpose: {fetch_bytes_x2, write_bytes, fabric_bytes}} - FETCH_SIZE (KiB) doubled per the gfx950 correction
calibrated on gathers of known bytes (profiles/r03_fetch_calibration.json), WRITE_SIZE exact."""
import csv
import json
import os
import sys


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mmssl_amd import graph, ops, synth
    d, U_r, I = 128, 250_000, 1_000_000
    raw = synth.interaction_matrix(U_r, I, 12_500_000, seed=11)
    P = graph.GraphPlan(synth.normalised_rows(raw))
    X = torch.randn(I, d, device="cuda")
    G = torch.randn(U_r, d, device="cuda")
    torch.cuda.synchronize()
    with torch.no_grad():
        for _ in range(3):
            ops.spmm(P, X)
        torch.cuda.synchronize()
        for _ in range(3):
            ops.spmm(P, G, transpose=True)
    torch.cuda.synchronize()


def rows(path, counter):
    out = []
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") == counter and "spmm_kernel" in r["Kernel_Name"]:
            out.append((int(r.get("Dispatch_Id", len(out))), float(r["Counter_Value"])))
    out.sort()
    return [v for _, v in out]


def summarise(fetch_csv, write_csv, out):
    f, w = rows(fetch_csv, "FETCH_SIZE"), rows(write_csv, "WRITE_SIZE")
    res = {"what": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (KiB, separate passes) over tools/spmm_hbm_pmc.py; launches 2-3 of "
                   "each flavour averaged; FETCH_SIZE x 2 (gfx950, calibrated: profiles/r03_fetch_calibration.json)",
           "level": "L2-miss / fabric requests: Infinity-Cache hits included (a 256 MiB cache in front of a 512 MB table still "
                    "catches the hot rows), so this is an upper bound of the HBM bytes"}
    for name, sl in (("forward", slice(1, 3)), ("transpose", slice(4, 6))):
        ff, ww = f[sl], w[sl]
        if not ff or not ww:
            continue
        fb, wb = 2 * 1024 * sum(ff) / len(ff), 1024 * sum(ww) / len(ww)
        res[name] = {"fetch_bytes_x2": int(fb), "write_bytes": int(wb), "fabric_bytes": int(fb + wb)}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res["source_sha256"] = bench.kernel_source_sha()      # bench.py withholds the figures when graph.hip has changed since
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "summarise":
        summarise(*sys.argv[2:5])
    else:
        run()
