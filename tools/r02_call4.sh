#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c4; mkdir -p $O
MMSSL_GEMM_V=7 timeout 300 python tools/gemm_v6_probe.py > $O/probe_v7.log 2>&1; echo "v7 rc=$?"; grep -v amdgpu $O/probe_v7.log | tail -8
MMSSL_GEMM_V=6 timeout 300 python tools/gemm_v6_probe.py > $O/probe_v6.log 2>&1; echo "v6 rc=$?"; grep -v amdgpu $O/probe_v6.log | tail -5
cp gpurun_out/gemm_v6_probe_*.json $O/
MMSSL_GEMM_V=7 timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "linear or Linear or projection" > $O/pytest_v7.log 2>&1; echo "pytest v7 rc=$?"; tail -3 $O/pytest_v7.log
MMSSL_GEMM_V=7 timeout 600 python bench.py --no-cpu-baseline --only steps > $O/bench_v7.json 2> $O/bench_v7.err; echo "bench v7 rc=$?"; cat $O/bench_v7.json
MMSSL_GEMM_V=6 timeout 600 python bench.py --no-cpu-baseline --only steps > $O/bench_v6.json 2> $O/bench_v6.err; echo "bench v6 rc=$?"; cat $O/bench_v6.json
MMSSL_GEMM_V=5 MMSSL_WGRAD_FT=0 timeout 600 python bench.py --no-cpu-baseline --only steps > $O/bench_v5.json 2> $O/bench_v5.err; echo "bench v5 rc=$?"; cat $O/bench_v5.json
cd /tmp
MMSSL_GEMM_V=7 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/trace_v7 -o t -- python /root/repo/bench.py --no-cpu-baseline --only steps --steps 30 --warmup 5 > /root/repo/$O/trace_v7.log 2>&1; echo "trace rc=$?"
cd /root/repo
python tools/trace_step.py $(find $O/trace_v7 -name '*kernel_trace.csv' | head -1) 12 --timeline > $O/step_timeline_v7.txt 2>&1
find $O -name '*_kernel_trace.csv' -size +30M -delete
