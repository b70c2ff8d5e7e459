#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c27; mkdir -p $O
tools/step_ab.sh $O/ab_wgrad.txt 4 "MMSSL_WGRAD_V=10" "MMSSL_WGRAD_V=5" | tail -3
