#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c17; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|^E " $O/pytest.log | head
for i in 1 2; do timeout 600 python bench.py --force-dist --steps 200 --warmup 20 > $O/bench_forcedist_$i.json 2> $O/bench_forcedist_$i.err; echo "forcedist rc=$?"; python -c "
import json
d=json.loads([l for l in open('$O/bench_forcedist_$i.json') if l.startswith('{')][0]); print('   ms_per_step', d['ms_per_step'], d['config']['launch'])"; done
timeout 600 python bench.py --no-cpu-baseline --only steps > $O/bench_n1.json 2>/dev/null; python -c "
import json
d=json.loads([l for l in open('$O/bench_n1.json') if l.startswith('{')][0]); print('   unsharded ms_per_step', d['ms_per_step'])"
