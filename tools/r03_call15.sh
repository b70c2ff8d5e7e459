#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c15; mkdir -p $O
R=$GRAFT_REPO_ROOT
MMSSL_TEST_KEEP=$O timeout 1800 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -4 $O/gpu_tests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err; python -c "
import json;b=json.load(open('$O/bench_driver_cmd.json'));print(b['ms_per_step'],b['value'],b['config']['spmm_launches_per_step'],b['config']['edge_layers_per_step'],b['loss_check']['rel_err']);print(b['roofline']);print(b['gcn_forward']['us'],b['gcn_forward']['frac_hbm']);print(b['projection'])"; tail -2 $O/bench.err
timeout 300 python bench.py --gpus 1 --steps 1000 --warmup 200 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('long run', b['ms_per_step'])"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/steps -o t -- python $R/bench.py --gpus 1 --steps 60 --warmup 20 --no-cpu-baseline --only steps > $R/$O/steps_bench.json 2>/dev/null
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/spmm_f -o f -- python $R/tools/spmm_pmc.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/spmm_w -o w -- python $R/tools/spmm_pmc.py > /dev/null 2>&1
cd $R
python tools/trace_step.py $O/steps/t_kernel_trace.csv 40 --timeline > $O/step_timeline.txt 2>&1; sed -n 26,70p $O/step_timeline.txt
python tools/pmc_summary.py $O/spmm_f/f_counter_collection.csv $O/spmm_w/w_counter_collection.csv $O/r03_spmm_pmc.json | tail -12
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +1M -delete
