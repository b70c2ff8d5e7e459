#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c2; mkdir -p $O
timeout 600 python -m pytest tests/test_proj_gpu.py -x -q > $O/proj_tests.log 2>&1; tail -15 $O/proj_tests.log
timeout 300 python tools/proj_probe.py > $O/proj_probe.json 2> $O/proj_probe.err; cat $O/proj_probe.json; tail -3 $O/proj_probe.err
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -x -q -k "g12 or 20_step" > $O/traj_tests.log 2>&1; tail -15 $O/traj_tests.log
