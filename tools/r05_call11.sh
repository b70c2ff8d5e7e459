#!/bin/bash
O=gpurun_out/r05k; mkdir -p $O
for R in 1 0 1 0; do
MMSSL_TEST_REPL=$R MMSSL_TEST_ROWTOL=1e-4 timeout 900 python -m pytest tests/test_synth_full_gpu.py -x -q -k "eight" > $O/synth8_$R.log 2>&1; echo "repl=$R rc=$?"; grep -o "AssertionError: ('E_[ui]', 'row-wise'.*amax', [0-9.e+-]*)" $O/synth8_$R.log | head -2
done
