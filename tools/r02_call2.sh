#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cat $O/bench_n1.json
timeout 600 python bench.py --force-dist --steps 100 --warmup 10 > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "forcedist rc=$?"; cat $O/bench_forcedist.json
MMSSL_DIST_STREAMS=0 timeout 600 python bench.py --force-dist --steps 100 --warmup 10 > $O/bench_forcedist_nostreams.json 2> $O/bench_forcedist_nostreams.err; echo "forcedist nostreams rc=$?"; cat $O/bench_forcedist_nostreams.json
timeout 600 python bench.py --force-dist --no-graph --steps 50 --warmup 5 > $O/bench_forcedist_eager.json 2> $O/bench_forcedist_eager.err; echo "forcedist eager rc=$?"; cat $O/bench_forcedist_eager.json
timeout 900 python bench.py --workload synth --steps 10 --warmup 2 > $O/bench_synth_w1.json 2> $O/bench_synth_w1.err; echo "synth rc=$?"; cat $O/bench_synth_w1.json
tail -3 $O/*.err
