#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c44; mkdir -p $O
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py tests/test_bench_contract_gpu.py tests/test_ops_gpu.py -x -q -k "not optin" 2>&1 | tail -2
tools/step_ab.sh $O/ab_ss.txt 3 "MMSSL_DEFER_SS=0" "MMSSL_DEFER_SS=1" | tail -2
