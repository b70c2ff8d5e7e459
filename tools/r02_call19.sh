#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c19; mkdir -p $O
for cfg in "MMSSL_GEMM_V=5" "MMSSL_GEMM_V=6" "MMSSL_GEMM_V=7 MMSSL_GEMM_PP_BK=32" "MMSSL_GEMM_V=7 MMSSL_GEMM_PP_BK=16" "MMSSL_GEMM_V=8" "MMSSL_GEMM_V=6 MMSSL_GEMM_SK_BLOCKS=768" "MMSSL_GEMM_V=6 MMSSL_GEMM_PRIO=1"; do
  env $cfg timeout 200 python tools/gemm_sustained_probe.py 2>&1 | grep -v amdgpu | tail -1 | sed "s/^/[$cfg] /" | tee -a $O/sustained.txt
done
