#!/bin/bash
O=gpurun_out/r05f; mkdir -p $O
timeout 600 python tools/step_probe.py 2>&1 | grep -v Warn | tee $O/step_probe.txt
timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_synth_full_gpu.py -k "not world_2_3_8 and not bench" > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -8 $O/gpu_tests.log
