#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c16; mkdir -p $O
MMSSL_GEMM_V=8 timeout 300 python tools/gemm_v6_probe.py > $O/probe_v8.log 2>&1; echo "v8 rc=$?"; grep -v amdgpu $O/probe_v8.log | tail -6
MMSSL_GEMM_V=6 timeout 300 python tools/gemm_v6_probe.py > $O/probe_v6.log 2>&1; echo "v6 rc=$?"; grep -v amdgpu $O/probe_v6.log | grep baby_img
MMSSL_GEMM_V=8 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "linear or Linear" > $O/pytest_v8.log 2>&1; echo "pytest v8 rc=$?"; tail -2 $O/pytest_v8.log
for cfg in "MMSSL_GEMM_V=8" "MMSSL_GEMM_V=8 MMSSL_WGRAD_FT=1" "MMSSL_GEMM_V=6"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg timeout 600 python bench.py --no-cpu-baseline --only steps > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench [$cfg] rc=$?"
  python -c "
import json
d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][0]); print('   ms_per_step', d['ms_per_step'], 'loss', d['config']['final_loss'])"
done
cp gpurun_out/gemm_v6_probe_V8*.json $O/ 2>/dev/null
