#!/bin/bash
# round 5, GPU call 5: full GPU suite on the split-precision default + bench lines
O=gpurun_out/r05e; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
tail -25 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --steps 500 --warmup 100 --no-cpu-baseline --no-hbm > $O/bench_split_$i.json 2> $O/bench.err
python - $i <<'PY'
import json, sys
d=json.loads([l for l in open("gpurun_out/r05e/bench_split_%s.json"%sys.argv[1]) if l.startswith("{")][0])
print("ms", d["ms_per_step"], "fwd", d["projection"]["forward"], "wgrad", d["projection"]["weight_gradient"], "roofline", d["roofline"]["frac"], "gcn", d["gcn_forward"]["frac_hbm"])
PY
done
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2>> $O/bench.err; head -c 600 $O/bench_driver_cmd.json; echo
