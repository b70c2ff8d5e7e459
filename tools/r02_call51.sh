#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c51; mkdir -p $O
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -2
tools/step_ab.sh $O/ab_parts.txt 3 "MMSSL_WGRAD_PARTS=0" "MMSSL_WGRAD_PARTS=1" | tail -2
MMSSL_WGRAD_PARTS=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof_steps -o s -- python /root/repo/bench.py --no-cpu-baseline --only steps --steps 50 --warmup 10 > /root/repo/$O/prof_steps.log 2>&1; echo "prof steps rc=$?"
python tools/trace_step.py $(find $O/prof_steps -name '*kernel_trace.csv' | head -1) 20 --timeline > $O/step_timeline.txt 2>&1
rm -f $O/prof_steps/*kernel_trace.csv $O/prof_steps/*/*kernel_trace.csv
