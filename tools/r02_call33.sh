#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c33; mkdir -p $O
timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "linear" 2>&1 | tail -1
MMSSL_WG10_DEPTH=16 timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k "linear" 2>&1 | tail -1
tools/step_ab.sh $O/ab_sched.txt 3 "MMSSL_BWD_SCHED=0" "MMSSL_BWD_SCHED=1" "MMSSL_BWD_SCHED=2" "MMSSL_WG10_DEPTH=16" "MMSSL_WG10_DEPTH=12" "MMSSL_WGRAD_V=5" | tail -6
for cfg in "MMSSL_WG10_DEPTH=16" "MMSSL_WG10_DEPTH=12"; do
  env $cfg PROBE_SECS=0.8 timeout 200 python tools/gemm_sustained_probe.py 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/[$cfg] /" | tee -a $O/sustained.txt
done
