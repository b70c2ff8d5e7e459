#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c62; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest_gpu.log
tools/step_ab.sh $O/ab_combine2.txt 2 "MMSSL_COMBINE2=0" "MMSSL_COMBINE2=1" | tail -2
