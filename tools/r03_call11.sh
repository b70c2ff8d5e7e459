#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c11; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('driver-cmd', b['ms_per_step'])"
timeout 300 python bench.py --gpus 1 --steps 1000 --warmup 200 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('long run', b['ms_per_step'])"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/steps -o t -- python $R/bench.py --gpus 1 --steps 60 --warmup 20 --no-cpu-baseline --only steps > $R/$O/steps_bench.json 2>/dev/null
cd $R
python tools/trace_step.py $O/steps/t_kernel_trace.csv 40 --timeline 2>&1 | sed -n 1,70p
find $O -name "*kernel_trace.csv" -delete
