#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c20; mkdir -p $O
MMSSL_GEMM_V=9 timeout 300 python -m pytest tests/test_ops_gpu.py -q -k "linear" > $O/pytest_v9.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_v9.log
for cfg in "MMSSL_GEMM_V=9" "MMSSL_GEMM_V=6"; do
  env $cfg timeout 200 python tools/gemm_sustained_probe.py 2>&1 | grep -v amdgpu | tail -3 | sed "s/^/[$cfg] /" | tee -a $O/sustained.txt
done
