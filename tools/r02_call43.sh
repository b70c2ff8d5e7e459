#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c43; mkdir -p $O
tools/step_ab.sh $O/ab_adamw.txt 3 "MMSSL_ADAMW_GROUPS=2" "MMSSL_ADAMW_GROUPS=1" | tail -2
