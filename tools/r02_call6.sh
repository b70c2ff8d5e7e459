#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c6; mkdir -p $O
MMSSL_GEMM_PP_BK=16 timeout 300 python tools/gemm_mode_probe.py > $O/mode_bk16.log 2>&1; echo "mode16 rc=$?"; grep "^mode" $O/mode_bk16.log
MMSSL_GEMM_PP_BK=32 timeout 300 python tools/gemm_mode_probe.py > $O/mode_bk32.log 2>&1; echo "mode32 rc=$?"; grep "^mode" $O/mode_bk32.log | head -3
MMSSL_GEMM_V=7 timeout 300 python tools/gemm_v6_probe.py > $O/probe_v7.log 2>&1; echo "v7 rc=$?"; grep -v amdgpu $O/probe_v7.log | tail -5
for cfg in "MMSSL_GEMM_V=7" "MMSSL_GEMM_V=6" "MMSSL_GEMM_V=6 MMSSL_LOSS_OVERLAP=1" "MMSSL_GEMM_V=6 MMSSL_COMBINE_FORK=1" "MMSSL_GEMM_V=5 MMSSL_WGRAD_FT=0"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg timeout 600 python bench.py --no-cpu-baseline --only steps > $O/bench_$tag.json 2> $O/bench_$tag.err; echo "bench [$cfg] rc=$?"
  python -c "
import json,sys
d=json.loads([l for l in open('$O/bench_$tag.json') if l.startswith('{')][0]); print('   ms_per_step', d['ms_per_step'], 'loss', d['config']['final_loss'])"
done
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
