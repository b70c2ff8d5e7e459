#!/bin/bash
O=gpurun_out/r05i; mkdir -p $O
timeout 1500 python -m pytest tests/test_dist_gpu.py -x -q -k "repl" > $O/dist_repl.log 2>&1; echo "dist repl rc=$?"; tail -4 $O/dist_repl.log
timeout 1500 python -m pytest tests/test_synth_full_gpu.py -x -q -k "eight" > $O/synth8.log 2>&1; echo "synth8 rc=$?"; tail -4 $O/synth8.log
timeout 900 python bench.py --gpus 8 --share-gpu --workload synth --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_synth_share8.json 2> $O/bench_synth_share8.err; echo "share8 rc=$?"
head -c 2500 $O/bench_synth_share8.json; tail -3 $O/bench_synth_share8.err
