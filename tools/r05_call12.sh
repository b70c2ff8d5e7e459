#!/bin/bash
O=gpurun_out/r05l; mkdir -p $O
timeout 900 python tools/spmm_stress.py 8 30 32 2>&1 | grep -v amdgpu.ids | tee $O/spmm_stress_8proc.txt
timeout 600 python tools/spmm_stress.py 1 30 32 2>&1 | grep -v amdgpu.ids | tee $O/spmm_stress_1proc.txt
