#!/bin/bash
# round 5, GPU call 2: split-precision projection v2 (A operand straight to registers): numerics, A/B, kernel trace
O=gpurun_out/r05b; mkdir -p $O
timeout 600 python -m pytest tests/test_proj_gpu.py -x -q > $O/proj_tests.log 2>&1; echo "proj tests rc=$?" | tee -a $O/summary.txt
tail -4 $O/proj_tests.log
for P in split f32; do
  timeout 300 python bench.py --steps 500 --warmup 100 --no-cpu-baseline --no-hbm --proj $P > $O/bench_$P.json 2> $O/bench_$P.err; echo "bench $P rc=$?" | tee -a $O/summary.txt
done
python - <<'PY'
import json
for n in ("split","f32"):
    try:
        d=json.loads([l for l in open("gpurun_out/r05b/bench_%s.json"%n) if l.startswith("{")][0])
        print(n, d["ms_per_step"], d["projection"]["forward"], d["projection"]["weight_gradient"], d["config"]["final_loss"])
    except Exception as e: print(n, "ERR", e)
PY
bash tools/evidence.sh r05b prof > $O/prof.log 2>&1
cp gpurun_out/r05bev/step_timeline.txt gpurun_out/r05bev/*_kernel_stats.csv $O/ 2>/dev/null
head -30 $O/r05b_rocprofv3_roofline_kernel_stats.csv
head -40 $O/step_timeline.txt
