#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c3; mkdir -p $O
MMSSL_GEMM_V=5 timeout 300 python tools/gemm_v6_probe.py > $O/probe_v5.log 2>&1; echo "v5 rc=$?"; grep -v amdgpu $O/probe_v5.log | tail -6
MMSSL_GEMM_V=6 timeout 300 python tools/gemm_v6_probe.py > $O/probe_v6.log 2>&1; echo "v6 rc=$?"; grep -v amdgpu $O/probe_v6.log | tail -6
MMSSL_GEMM_V=6 MMSSL_GEMM_SK_BLOCKS=768 timeout 300 python tools/gemm_v6_probe.py > $O/probe_v6_768.log 2>&1; echo "v6/768 rc=$?"; grep -v amdgpu $O/probe_v6_768.log | tail -6
MMSSL_GEMM_V=6 MMSSL_GEMM_SK_BLOCKS=256 timeout 300 python tools/gemm_v6_probe.py > $O/probe_v6_256.log 2>&1; echo "v6/256 rc=$?"; grep -v amdgpu $O/probe_v6_256.log | tail -6
cp gpurun_out/gemm_v6_probe_*.json $O/
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cat $O/bench_n1.json
timeout 600 python bench.py --force-dist --steps 100 --warmup 10 > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "forcedist rc=$?"; cat $O/bench_forcedist.json
timeout 900 python bench.py --workload synth --steps 10 --warmup 2 > $O/bench_synth_w1.json 2> $O/bench_synth_w1.err; echo "synth rc=$?"; cat $O/bench_synth_w1.json
