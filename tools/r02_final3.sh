#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02final3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cut -c1-260 $O/bench_n1.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_steps -o s -- python /root/repo/bench.py --no-cpu-baseline --only steps --steps 50 --warmup 10 > /root/repo/$O/prof_steps.log 2>&1; echo "prof steps rc=$?"
cd /root/repo
python tools/trace_step.py $(find $O/prof_steps -name '*kernel_trace.csv' | head -1) 20 --timeline > $O/step_timeline.txt 2>&1
rm -f $O/prof_steps/*kernel_trace.csv
