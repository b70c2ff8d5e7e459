#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c37; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
tools/step_ab.sh $O/ab_fixup.txt 3 "MMSSL_GEMM_FIXUP=0" "MMSSL_GEMM_FIXUP=1" | tail -2
for cfg in "MMSSL_GEMM_FIXUP=1" "MMSSL_GEMM_FIXUP=0"; do
  env $cfg PROBE_SECS=0.8 timeout 200 python tools/gemm_sustained_probe.py 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/[$cfg] /" | tee -a $O/sustained.txt
done
