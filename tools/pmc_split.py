"""Per-kernel counter averages from a rocprofv3 counter_collection.csv, split into consecutive GROUPS of dispatches
(the per-modality workload of rounds 1-3, tools/gemm_pmc.py - removed from the tree in round 5, in git history - launched 6 x the image shape, then 6 x the text shape; tools/proj_pmc.py launches one grouped shape): python tools/pmc_split.py CSV 6 [filter]"""
import collections
import csv
import sys

path, group = sys.argv[1], int(sys.argv[2])
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = collections.defaultdict(lambda: collections.defaultdict(dict))     # kernel -> dispatch -> counter -> value
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    if flt not in k or "at::native" in k:
        continue
    d = int(r["Dispatch_Id"])
    rows[k][d][r["Counter_Name"]] = float(r["Counter_Value"])
    dur[k][d] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, disp in rows.items():
    ids = sorted(disp)
    for g in range(0, len(ids), group):
        sel = ids[g:g + group][1:] or ids[g:g + group]          # drop the group's first (cold) launch
        print("%s   dispatches %d..%d  avg %.1f us (under the profiler)" % (k, sel[0], sel[-1], sum(dur[k][i] for i in sel) / len(sel)))
        for c in sorted(disp[sel[0]]):
            print("   %-30s %14.1f" % (c, sum(disp[i][c] for i in sel) / len(sel)))
