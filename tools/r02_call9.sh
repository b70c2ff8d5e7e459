#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c9; mkdir -p $O
timeout 900 python tools/split_debug.py > $O/split_debug.log 2>&1; echo "split rc=$?"; grep -E "^===|warm-up|OK|rror|Segmentation" $O/split_debug.log
timeout 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k "three_modalities" > $O/pytest_j1.log 2>&1; echo "pytest j1 rc=$?"; tail -5 $O/pytest_j1.log
