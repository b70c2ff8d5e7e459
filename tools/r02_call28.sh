#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c28; mkdir -p $O
MMSSL_GEMM_NT=1 timeout 300 python -m pytest tests/test_ops_gpu.py -q -x -k linear 2>&1 | tail -1
tools/step_ab.sh $O/ab_nt.txt 3 "MMSSL_GEMM_NT=0" "MMSSL_GEMM_NT=1" | tail -3
for cfg in "MMSSL_GEMM_NT=1"; do
  env $cfg PROBE_SECS=0.8 timeout 200 python tools/gemm_sustained_probe.py 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/[$cfg] /" | tee -a $O/sustained.txt
done
