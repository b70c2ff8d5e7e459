#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c23; mkdir -p $O
for cfg in "MMSSL_GEMM_V=9" "MMSSL_GEMM_V=9 MMSSL_GEMM_R9_MODE=4" "MMSSL_GEMM_V=9 MMSSL_GEMM_R9_MODE=6" "MMSSL_GEMM_V=9 MMSSL_GEMM_R9_MODE=2" "MMSSL_GEMM_V=9 MMSSL_GEMM_R9_MODE=1"; do
  env $cfg timeout 200 python tools/gemm_sustained_probe.py 2>&1 | grep -v amdgpu | tail -3 | sed "s/^/[$cfg] /" | tee -a $O/sustained.txt
done
cd /tmp; MMSSL_GEMM_V=9 rocprofv3 --kernel-trace --stats -d /tmp/r9prof -o r9 -- python $GRAFT_REPO_ROOT/tools/gemm_sustained_probe.py > /dev/null 2>&1
cp /tmp/r9prof/*kernel_stats.csv $GRAFT_REPO_ROOT/$O/ 2>/dev/null; head -8 /tmp/r9prof/*kernel_stats.csv
