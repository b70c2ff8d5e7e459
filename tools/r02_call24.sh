#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c24; mkdir -p $O
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x > $O/pytest_ops.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_ops.log
for cfg in "MMSSL_WGRAD_V=10" "MMSSL_WGRAD_V=10 MMSSL_WG10_BLOCKS=256" "MMSSL_WGRAD_V=10 MMSSL_WG10_BLOCKS=1024" "MMSSL_WGRAD_V=5"; do
  env $cfg timeout 200 python tools/gemm_sustained_probe.py 2>&1 | grep -v amdgpu | tail -3 | sed "s/^/[$cfg] /" | tee -a $O/sustained.txt
done
