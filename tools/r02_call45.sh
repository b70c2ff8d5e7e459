#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c45; mkdir -p $O
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py tests/test_bench_contract_gpu.py -x -q 2>&1 | tail -2
tools/step_ab.sh $O/ab_bubbles.txt 3 "MMSSL_PREFILL=0 MMSSL_LAZY_ANCHOR=0" "MMSSL_PREFILL=1 MMSSL_LAZY_ANCHOR=0" "MMSSL_PREFILL=0 MMSSL_LAZY_ANCHOR=1" "MMSSL_PREFILL=1 MMSSL_LAZY_ANCHOR=1" | tail -4
