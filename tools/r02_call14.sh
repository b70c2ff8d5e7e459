#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c14; mkdir -p $O
timeout 300 python tools/gemm_prio_probe.py > $O/prio.log 2>&1; echo "prio rc=$?"; grep -v amdgpu $O/prio.log
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -q -k "device_modal or trainer" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED|^E " $O/pytest.log | head
timeout 600 python tools/trainer_bench.py --workload baby --batches 10 > $O/trainer_bench.log 2>&1; echo "trainer bench rc=$?"; grep -v amdgpu $O/trainer_bench.log | tail -1
timeout 600 python bench.py --force-dist --steps 100 --warmup 10 > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "forcedist rc=$?"; python -c "
import json
d=json.loads([l for l in open('$O/bench_forcedist.json') if l.startswith('{')][0]); print('   ms_per_step', d['ms_per_step'], d['config']['launch'], d['comm'])"
