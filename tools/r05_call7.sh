#!/bin/bash
O=gpurun_out/r05g; mkdir -p $O
timeout 2000 python -m pytest tests/test_synth_full_gpu.py -x -q --durations=5 > $O/synth_full_tests.log 2>&1; echo "synth full tests rc=$?"
tail -15 $O/synth_full_tests.log
