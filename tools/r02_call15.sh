#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c15; mkdir -p $O
timeout 600 python bench.py --force-dist --steps 100 --warmup 10 > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "forcedist rc=$?"; python -c "
import json
d=json.loads([l for l in open('$O/bench_forcedist.json') if l.startswith('{')][0]); print('   ms_per_step', d['ms_per_step'], d['config']['launch'])"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/trace_fd -o t -- python /root/repo/bench.py --force-dist --steps 30 --warmup 5 > /root/repo/$O/trace_fd.log 2>&1; echo "trace rc=$?"
cd /root/repo
python tools/trace_step.py $(find $O/trace_fd -name '*kernel_trace.csv' | head -1) 12 --timeline > $O/step_timeline_forcedist.txt 2>&1
head -60 $O/step_timeline_forcedist.txt
find $O -name '*_kernel_trace.csv' -size +30M -delete
