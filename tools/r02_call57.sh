#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c57; mkdir -p $O
tools/step_ab.sh $O/ab_prio.txt 2 "MMSSL_STREAM_PRIO=0,0,0" "MMSSL_STREAM_PRIO=0,0,-1" "MMSSL_STREAM_PRIO=-1,-1,0" "MMSSL_STREAM_PRIO=-1,0,0" | tail -4
