#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c58; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/prof -o s -- python /root/repo/bench.py --force-dist --no-cpu-baseline --only steps --steps 50 --warmup 10 > /root/repo/$O/prof.log 2>&1; echo "prof rc=$?"
python tools/trace_step.py $(find $O/prof -name '*kernel_trace.csv' | head -1) 20 --timeline --marker=prep_kernel > $O/step_timeline_forcedist.txt 2>&1
rm -f $O/prof/*kernel_trace.csv $O/prof/*/*kernel_trace.csv
