"""Summarise rocprofv3 counter_collection CSVs of tools/spmm_pmc.py into profiles/spmm_pmc.json.
FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests at 64 B, so it is doubled
(guides/MI355X_MICROARCH.md, section HBM, calibrated there for wide streaming reads; profiles/r03_fetch_calibration.json
calibrates it for THIS kernel's 256-B / 512-B row gathers on a gather of known size: ratio 2.00 / 1.97; WRITE_SIZE exact)."""
import csv
import json
import sys


def per_kernel(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"]
        a = acc.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in acc.items()}, {k: v[0] for k, v in acc.items()}


def main(fetch_csv, write_csv, out):
    f, nf = per_kernel(fetch_csv, "FETCH_SIZE")
    w, nw = per_kernel(write_csv, "WRITE_SIZE")
    res = {"kernels": {}}
    tot_f = tot_w = n_main = 0
    for k in f:
        short = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        if "spmm" not in short:
            continue
        res["kernels"][short] = {"launches": nf[k], "FETCH_SIZE_KiB_avg": round(f[k], 1),
                                 "WRITE_SIZE_KiB_avg": round(w.get(k, 0.0), 1)}
        tot_f += f[k] * nf[k]
        tot_w += w.get(k, 0.0) * nw.get(k, 0)
        if "spmm_kernel" in short:
            n_main += nf[k]
    # one "SpMM launch" = main kernel + its second-stage kernel
    res["fetch_bytes_per_launch_raw"] = int(tot_f * 1024 / n_main)
    res["fetch_bytes_per_launch_x2_gfx950"] = int(2 * tot_f * 1024 / n_main)
    res["write_bytes_per_launch"] = int(tot_w * 1024 / n_main)
    res["hbm_bytes_per_launch"] = res["fetch_bytes_per_launch_x2_gfx950"] + res["write_bytes_per_launch"]
    res["level"] = "L2-miss / fabric requests (TCC_EA0): Infinity-Cache hits are included, so this is NOT an HBM byte count"
    res["note"] = ("tables and CSR of the Baby shape are Infinity-Cache resident: this fabric traffic is below the algorithmic "
                   "gather-per-edge bytes (74.7 MB/launch at d = 64) but above the compulsory once-through bytes (~16 MB) - "
                   "every XCD's L2 pulls its own copy of the gathered table")
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    res["source_sha256"] = bench.kernel_source_sha()      # bench.py withholds `traffic` when graph.hip has changed since
    res["source_files"] = list(bench.SPMM_SOURCES)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
