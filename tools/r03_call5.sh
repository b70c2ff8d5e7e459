#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c5; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -25 $O/gpu_tests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err; python -c "
import json;b=json.load(open('$O/bench_driver_cmd.json'));print(b['ms_per_step'],b['value'],b['config']['spmm_launches_per_step'],b['config']['launch'],b.get('loss_check'))"; tail -3 $O/bench.err
timeout 300 python bench.py --gpus 1 --steps 1000 --warmup 200 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('long run', b['ms_per_step'])"
