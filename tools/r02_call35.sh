#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c35; mkdir -p $O
tools/step_ab.sh $O/ab_blocks.txt 3 "MMSSL_WG10_BLOCKS=256" "MMSSL_WG10_BLOCKS=512" "MMSSL_WG10_BLOCKS=1024" "MMSSL_WGRAD_V=5" | tail -4
