#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c18; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_proj_gpu.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "ring or hotpath or g12 or trajectory" 2>&1 | tail -3
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err; python -c "
import json;b=json.load(open('$O/bench_driver_cmd.json'));print(b['ms_per_step'],b['value'],b['config']['final_loss'],b['loss_check']['rel_err']);print(b['projection'])"
for fl in "" "--two-launch-proj"; do
timeout 300 python bench.py --gpus 1 --steps 1000 --warmup 200 --no-cpu-baseline --only steps $fl 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('long run $fl', b['ms_per_step'], b['config']['final_loss'])"
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/steps -o t -- python $R/bench.py --gpus 1 --steps 60 --warmup 20 --no-cpu-baseline --only steps > $R/$O/steps_bench.json 2>/dev/null
cd $R
python tools/trace_step.py $O/steps/t_kernel_trace.csv 40 --timeline > $O/step_timeline.txt 2>&1; cat $O/step_timeline.txt | tail -36
find $O -name "*kernel_trace.csv" -delete
