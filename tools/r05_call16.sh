#!/bin/bash
O=gpurun_out/r05n; mkdir -p $O
timeout 600 python -m pytest tests/test_proj_gpu.py -x -q > $O/proj_tests.log 2>&1; echo "proj tests rc=$?"; tail -3 $O/proj_tests.log
for i in 1 2; do
timeout 300 python bench.py --steps 500 --warmup 100 --no-cpu-baseline --no-hbm --no-frows > $O/bench_$i.json 2> $O/bench.err
python - $i <<'PY'
import json, sys
d=json.loads([l for l in open("gpurun_out/r05n/bench_%s.json"%sys.argv[1]) if l.startswith("{")][0])
print("ms", d["ms_per_step"], "fwd", d["projection"]["forward"], "wgrad", d["projection"]["weight_gradient"])
PY
done
