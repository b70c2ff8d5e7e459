#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c10; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_proj_gpu.py -x -q 2>&1 | tail -3
timeout 120 python tools/proj_probe.py --only-new --secs 0.5 2>/dev/null
for d in 4 5 6; do echo "dbg $d"; MMSSL_LIB=$R/tools/_dbg/libmmssl_dbg$d.so timeout 120 python tools/proj_probe.py --only-new --secs 0.3 2>/dev/null; done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('driver-cmd', b['ms_per_step'])"
timeout 300 python bench.py --gpus 1 --steps 1000 --warmup 200 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('long run', b['ms_per_step'])"
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py tests/test_ops_gpu.py -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
