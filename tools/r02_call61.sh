#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c61; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py -x -q -k "not optin" 2>&1 | tail -2
tools/step_ab.sh $O/ab_guest.txt 3 "MMSSL_BPR_GUEST=0" "MMSSL_BPR_GUEST=1" | tail -2
