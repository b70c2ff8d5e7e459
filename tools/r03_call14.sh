#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_proj_gpu.py -x -q 2>&1 | tail -1
timeout 120 python tools/proj_probe.py --only-new --secs 0.5 2>/dev/null
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('driver-cmd', b['ms_per_step'])"
timeout 300 python bench.py --gpus 1 --steps 1000 --warmup 200 --no-cpu-baseline --only steps 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('long run', b['ms_per_step'])"
