#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c56; mkdir -p $O
tools/step_ab.sh $O/ab_fwdorder.txt 2 "MMSSL_FWD_ORDER=ABC" "MMSSL_FWD_ORDER=BAC" "MMSSL_FWD_ORDER=CAB" "MMSSL_FWD_ORDER=ACB" "MMSSL_BWD_ORDER=ABC" "MMSSL_BWD_ORDER=ACB" | tail -6
