#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c7; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "hot_node or hotpath or baby_full or g12 or 20_step or trainer" > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
for w in 4 8; do
  MMSSL_PROJ_WAVES=$w timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('waves $w driver-cmd', b['ms_per_step'])"
  MMSSL_PROJ_WAVES=$w timeout 300 python bench.py --gpus 1 --steps 1000 --warmup 200 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('waves $w long run', b['ms_per_step'])"
done
cd /tmp
MMSSL_PROJ_WAVES=8 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/steps -o t -- python $R/bench.py --gpus 1 --steps 60 --warmup 20 --no-cpu-baseline --only steps > $R/$O/steps_bench.json 2>/dev/null
cd $R
python tools/trace_step.py $O/steps/t_kernel_trace.csv 40 --timeline > $O/step_timeline.txt 2>&1; cat $O/step_timeline.txt
find $O -name "*kernel_trace.csv" -delete
