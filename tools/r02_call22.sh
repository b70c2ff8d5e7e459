#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c22; mkdir -p $O
for cfg in "MMSSL_GEMM_V=9" "MMSSL_GEMM_V=6" "MMSSL_GEMM_V=9 MMSSL_GEMM_R9_MODE=2"; do
  env $cfg timeout 200 python tools/gemm_stride_probe.py 2>&1 | grep -v amdgpu | sed "s/^/[$cfg] /" | tee -a $O/stride.txt
done
