#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c41; mkdir -p $O
for cfg in "PROBE_FT_FWD=1" "PROBE_FT_FWD=1 MMSSL_FT_SPLITS=4" "PROBE_FT_FWD=1 MMSSL_FT_SPLITS=2" "PROBE_FT_FWD=1 MMSSL_FT_SPLITS=7"; do
  env $cfg PROBE_SECS=0.8 timeout 200 python tools/gemm_sustained_probe.py 2>&1 | grep -v amdgpu | tail -4 | sed "s/^/[$cfg] /" | tee -a $O/sustained.txt
done
