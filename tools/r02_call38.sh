#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c38; mkdir -p $O
timeout 600 python tools/step_power_probe.py 2>&1 | grep -v amdgpu | tee $O/step_power.txt
