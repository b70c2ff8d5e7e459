#!/bin/bash
export TMPDIR=/tmp
O=$PWD/gpurun_out/r02c25; mkdir -p $O
cd /tmp
PROBE_SECS=0.15 MMSSL_WG10_BLOCKS=256 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wgprof -o w -- python $GRAFT_REPO_ROOT/tools/gemm_sustained_probe.py > $O/probe.log 2>&1
f=$(find /tmp/wgprof -name "*kernel_stats.csv" | head -1); echo "stats file: $f"
cp "$f" $O/wg10_kernel_stats.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %6s avg %8.1f us min %8.1f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
