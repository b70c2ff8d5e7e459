#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c53; mkdir -p $O
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_ops_gpu.py -x -q -k "baby or eager_chain" 2>&1 | tail -1
tools/step_ab.sh $O/ab_maxsp.txt 3 "MMSSL_WG10_MAXSP=8" "MMSSL_WG10_MAXSP=4" "MMSSL_WG10_MAXSP=16" "MMSSL_WG10_MAXSP=8 MMSSL_WG10_BLOCKS=256" | tail -4
