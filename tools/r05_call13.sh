#!/bin/bash
O=gpurun_out/r05m; mkdir -p $O
MMSSL_TEST_REPEAT=1 MMSSL_TEST_ROWTOL=1 timeout 1200 python -m pytest tests/test_synth_full_gpu.py -x -q -s -k "eight" > $O/synth8_repeat.log 2>&1; echo "rc=$?"; grep REPEAT $O/synth8_repeat.log | cut -c1-900
