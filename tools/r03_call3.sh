#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c3; mkdir -p $O
for w in 4 8; do
  MMSSL_PROJ_WAVES=$w timeout 600 python -m pytest tests/test_proj_gpu.py -x -q 2>&1 | tail -3
  MMSSL_PROJ_WAVES=$w timeout 300 python tools/proj_probe.py > $O/proj_probe_w$w.json 2> $O/proj_probe.err; cat $O/proj_probe_w$w.json
done
cd /tmp
MMSSL_PROJ_WAVES=4 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace4 -o t -- python $GRAFT_REPO_ROOT/tools/proj_probe.py --secs 0.2 > /dev/null 2>&1
MMSSL_PROJ_WAVES=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace8 -o t -- python $GRAFT_REPO_ROOT/tools/proj_probe.py --secs 0.2 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for w in 4 8; do echo "== waves $w"; python - <<PY
import csv
rows=list(csv.DictReader(open("$O/trace$w/t_kernel_stats.csv")))
for r in rows[:12]:
    print("%-60s calls %6s avg %9.1f us  tot%% %s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
done
find $O -name "*trace.csv" -delete
