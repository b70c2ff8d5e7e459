#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c18; mkdir -p $O
timeout 600 python tools/gemm_data_probe.py > $O/data_probe.log 2>&1; echo "rc=$?"; grep -v amdgpu $O/data_probe.log
