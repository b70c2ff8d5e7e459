#!/bin/bash
# one captured step's kernel timeline (rocprofv3 --kernel-trace of the driver command, steps only): bash tools/step_timeline.sh NAME
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/${1:-timeline}
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/steps -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --only steps > $R/$O/bench.json 2> $R/$O/bench.err
cd $R
python tools/trace_step.py $O/steps/t_kernel_trace.csv 12 --timeline > $O/step_timeline.txt 2>&1
rm -rf $O/steps
