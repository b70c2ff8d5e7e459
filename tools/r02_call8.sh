#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c8; mkdir -p $O
timeout 900 python tools/split_debug.py > $O/split_debug.log 2>&1; echo "split rc=$?"; grep -E "^===|capture returned|OK|Error|error|Segmentation|capturing" $O/split_debug.log
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > $O/pytest_ops.log 2>&1; echo "pytest ops rc=$?"; tail -5 $O/pytest_ops.log
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "eval or usim or generator_step" > $O/pytest_model.log 2>&1; echo "pytest model rc=$?"; tail -5 $O/pytest_model.log
timeout 600 python bench.py --no-cpu-baseline --only steps > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python -c "
import json
d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][0]); print('   ms_per_step', d['ms_per_step'])"
