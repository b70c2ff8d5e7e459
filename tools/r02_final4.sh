#!/bin/bash
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python -m pytest tests/test_fullsize_gpu.py -q -k "baby_full_step" 2>&1 | tail -1
timeout 120 python bench.py --no-cpu-baseline --only steps 2>/dev/null | tail -1 | cut -c1-220
