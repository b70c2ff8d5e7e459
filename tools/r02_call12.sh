#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c12; mkdir -p $O
timeout 900 python tools/split_debug.py > $O/split_debug.log 2>&1; echo "split rc=$?"; grep -E "^===|capture returned|warm-up|OK|Segmentation" $O/split_debug.log
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Segmentation|^E " $O/pytest.log | head -20
