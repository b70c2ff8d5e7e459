#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c26; mkdir -p $O
for cfg in "MMSSL_WGRAD_V=10" "MMSSL_WGRAD_V=5" "MMSSL_WGRAD_V=10" "MMSSL_WGRAD_V=5"; do
  env $cfg timeout 300 python bench.py --no-cpu-baseline --only steps --steps 300 --warmup 30 2>&1 | grep -v amdgpu | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('$cfg', r['ms_per_step'], r['value'])" | tee -a $O/bench_ab.txt
done
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
