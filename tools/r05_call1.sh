#!/bin/bash
# round 5, GPU call 1: split-precision projection numerics + A/B, configs[4] at full size
O=gpurun_out/r05a; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_proj_gpu.py -x -q > $O/proj_tests.log 2>&1; echo "proj tests rc=$?" | tee -a $O/summary.txt
tail -5 $O/proj_tests.log
timeout 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-hbm > $O/bench_split.json 2> $O/bench_split.err; echo "bench split rc=$?" | tee -a $O/summary.txt
timeout 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-hbm --proj f32 > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json
for n in ("split","f32"):
    try:
        d=json.loads([l for l in open("gpurun_out/r05a/bench_%s.json"%n) if l.startswith("{")][0])
        print(n, d["ms_per_step"], d["projection"]["forward"], d["projection"]["weight_gradient"], d["config"]["final_loss"])
    except Exception as e: print(n, "ERR", e)
PY
timeout 1500 python -m pytest tests/test_synth_full_gpu.py -x -q --durations=5 > $O/synth_full_tests.log 2>&1; echo "synth full tests rc=$?" | tee -a $O/summary.txt
tail -15 $O/synth_full_tests.log
timeout 600 python bench.py --workload synth-full --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_synth_full_n1.json 2> $O/bench_synth_full_n1.err; echo "bench synth-full rc=$?" | tee -a $O/summary.txt
head -c 1500 $O/bench_synth_full_n1.json; tail -3 $O/bench_synth_full_n1.err
