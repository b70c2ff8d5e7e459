#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02final; mkdir -p $O
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cut -c1-300 $O/bench_n1.json
timeout 600 python bench.py --force-dist > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "forcedist rc=$?"; head -1 $O/bench_forcedist.json | cut -c1-300
