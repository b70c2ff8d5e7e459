#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c46; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py -x -q 2>&1 | tail -2
tools/step_ab.sh $O/ab_wgfix.txt 3 "MMSSL_WGRAD_FIXUP=0" "MMSSL_WGRAD_FIXUP=1" | tail -2
for cfg in "MMSSL_WGRAD_FIXUP=1" "MMSSL_WGRAD_FIXUP=0"; do
  env $cfg PROBE_SECS=0.8 timeout 200 python tools/gemm_sustained_probe.py 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/[$cfg] /" | cut -c1-200 | tee -a $O/sustained.txt
done
