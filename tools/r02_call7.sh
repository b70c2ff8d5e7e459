#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c7; mkdir -p $O
timeout 900 python tools/split_debug.py > $O/split_debug.log 2>&1; echo "split rc=$?"; cat $O/split_debug.log
