#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
