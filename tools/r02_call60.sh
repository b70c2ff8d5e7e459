#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c60; mkdir -p $O
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py tests/test_bench_contract_gpu.py -x -q 2>&1 | tail -2
tools/step_ab.sh $O/ab_maskahead.txt 3 "MMSSL_MASK_AHEAD=0" "MMSSL_MASK_AHEAD=1" | tail -2
