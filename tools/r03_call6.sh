#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c6; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q --deselect tests/test_dist_nccl_gpu.py > $O/gpu_tests.log 2>&1; tail -8 $O/gpu_tests.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/steps -o t -- python $R/bench.py --gpus 1 --steps 60 --warmup 20 --no-cpu-baseline --only steps > $R/$O/steps_bench.json 2>/dev/null
cd $R
python tools/trace_step.py $O/steps/t_kernel_trace.csv 40 --timeline > $O/step_timeline.txt 2>&1; cat $O/step_timeline.txt
cat $O/steps_bench.json | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"
find $O -name "*kernel_trace.csv" -delete
