#!/bin/bash
# the round's closing evidence on the final build, in one gpurun call (bash tools/r03_final.sh): GPU suite, smoke, the
# driver's bench command, the default bench, the sharded code path on one rank, rocprofv3 kernel stats + one step's timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/final
mkdir -p $R/$O
cd $R
python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" >> $O/gpu_tests.log 2>&1
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err
timeout 300 python bench.py > $O/bench_n1.json 2>> $O/bench.err
timeout 300 python bench.py --gpus 1 --steps 500 --warmup 100 --no-cpu-baseline --only steps --force-dist > $O/bench_forcedist.json 2>> $O/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/steps -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --only steps > /dev/null 2>&1
cd $R
python tools/trace_step.py $O/steps/t_kernel_trace.csv 12 --timeline > $O/step_timeline.txt 2>&1
cp $O/steps/t_kernel_stats.csv $O/steps_kernel_stats.csv 2>/dev/null || find $O/steps -name "*kernel_stats.csv" -exec cp {} $O/steps_kernel_stats.csv \;
rm -rf $O/steps
cat $O/gpu_tests.log
for f in bench_driver_cmd bench_n1 bench_forcedist; do python -c "
import json; b=json.load(open('$O/$f.json')); print('$f', b['ms_per_step'], b['value'], b.get('roofline',{}).get('frac'), b.get('projection',{}).get('forward'), b.get('projection',{}).get('weight_gradient'))"; done
