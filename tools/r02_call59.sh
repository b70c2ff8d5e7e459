#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c59; mkdir -p $O
timeout 600 python -m pytest tests/test_dist_gpu.py tests/test_bench_contract_gpu.py -x -q 2>&1 | tail -3
tools/step_ab.sh $O/ab_fd.txt 2 "X=1" | tail -1
for i in 1 2; do timeout 300 python bench.py --force-dist --no-cpu-baseline 2>/dev/null | head -1 | python -c "import sys,json; print('force-dist', json.loads(sys.stdin.readline())['ms_per_step'])"; done
