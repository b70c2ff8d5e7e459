#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c34; mkdir -p $O
timeout 600 python tools/recapture_probe.py 2>&1 | grep -v amdgpu | tee $O/recapture.txt
timeout 600 python tools/recapture_probe.py 2>&1 | grep -v amdgpu | tee -a $O/recapture.txt
