#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c54; mkdir -p $O
tools/step_ab.sh $O/ab_skblocks.txt 3 "MMSSL_GEMM_SK_BLOCKS=512" "MMSSL_GEMM_SK_BLOCKS=384" "MMSSL_GEMM_SK_BLOCKS=256" "MMSSL_GEMM_SK_BLOCKS=768" | tail -4
