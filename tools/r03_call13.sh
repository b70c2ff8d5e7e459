#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c13; mkdir -p $O
timeout 600 python -m pytest tests/test_proj_gpu.py -x -q 2>&1 | tail -2
timeout 120 python tools/proj_probe.py --only-new --secs 0.5 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('driver-cmd', b['ms_per_step'])"
timeout 300 python bench.py --gpus 1 --steps 1000 --warmup 200 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('long run', b['ms_per_step'])"
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_model_gpu.py tests/test_ops_gpu.py -m gpu -x -q -k "hot or baby or 20_step or g12 or proj" 2>&1 | tail -2
