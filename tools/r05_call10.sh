#!/bin/bash
O=gpurun_out/r05j; mkdir -p $O
for D in 4 6 8; do
  MMSSL_PROJX_DEPTH=$D timeout 300 python -m pytest tests/test_proj_gpu.py -x -q -k "split" > $O/proj_tests_d$D.log 2>&1; echo "depth $D tests rc=$?"
  MMSSL_PROJX_DEPTH=$D timeout 300 python bench.py --steps 500 --warmup 100 --no-cpu-baseline --no-hbm --no-frows > $O/bench_d$D.json 2> $O/bench_d$D.err
  python - $D <<'PY'
import json, sys
b=sys.argv[1]
try:
    d=json.loads([l for l in open("gpurun_out/r05j/bench_d%s.json"%b) if l.startswith("{")][0])
    print("depth", b, "ms", d["ms_per_step"], "fwd", d["projection"]["forward"]["us"], "wgrad", d["projection"]["weight_gradient"]["us"])
except Exception as e: print(b, "ERR", e)
PY
done
timeout 1500 python -m pytest tests/test_synth_full_gpu.py -x -q -k "eight" > $O/synth8.log 2>&1; echo "synth8 rc=$?"; tail -3 $O/synth8.log
