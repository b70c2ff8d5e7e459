#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c13; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|Segmentation|^E " $O/pytest.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 600 python tools/trainer_bench.py --workload baby --batches 10 > $O/trainer_bench.log 2>&1; echo "trainer bench rc=$?"; grep -v amdgpu $O/trainer_bench.log | tail -12
