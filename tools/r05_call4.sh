#!/bin/bash
# round 5, GPU call 4: projx block-count sweep (step time, un-profiled) + configs[4] tests against the float64-accumulated golden
O=gpurun_out/r05d; mkdir -p $O
for B in 0 240 224 208 192 160; do
  MMSSL_PROJX_BLOCKS=$B timeout 300 python bench.py --steps 500 --warmup 100 --no-cpu-baseline --no-hbm > $O/bench_b$B.json 2> $O/bench_b$B.err
  python - $B <<'PY'
import json, sys
b=sys.argv[1]
try:
    d=json.loads([l for l in open("gpurun_out/r05d/bench_b%s.json"%b) if l.startswith("{")][0])
    print("blocks", b, "ms", d["ms_per_step"], "fwd", d["projection"]["forward"]["us"], "wgrad", d["projection"]["weight_gradient"]["us"])
except Exception as e: print(b, "ERR", e)
PY
done
if [ -f tests/golden/synth_full_n1.npz ]; then
  timeout 1500 python -m pytest tests/test_synth_full_gpu.py -x -q --durations=5 > $O/synth_full_tests.log 2>&1; echo "synth full tests rc=$?"
  tail -12 $O/synth_full_tests.log
fi
