cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "loss or hotpath or trajectory or g12 or hot_node or infonce or bpr" 2>&1 | tail -6
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('driver cmd', b['ms_per_step'], b['config']['final_loss'], b['loss_check'])"
for i in 1 2; do
timeout 300 python bench.py --gpus 1 --steps 1000 --warmup 200 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('long', b['ms_per_step'], b['config']['final_loss'])"
done
cd /tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r03c30; mkdir -p $R/$O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/steps -o t -- python $R/bench.py --gpus 1 --steps 60 --warmup 20 --no-cpu-baseline --only steps > /dev/null 2>&1
cd $R
python tools/trace_step.py $O/steps/t_kernel_trace.csv 40 --timeline > $O/step_timeline.txt 2>&1; sed -n '/^timeline/,$p' $O/step_timeline.txt | head -12
find $O -name "*kernel_trace.csv" -delete
