#!/bin/bash
# round-2 evidence run: bench lines, separate rocprofv3 stats for the steps and for the roofline loop, step timeline,
# sharded path on one rank, configs[4] per-rank share, kbench, trainer loop
export TMPDIR=/tmp
O=gpurun_out/r02ev; mkdir -p $O
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?"; cat $O/bench_n1.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_steps -o s -- python /root/repo/bench.py --no-cpu-baseline --only steps --steps 50 --warmup 10 > /root/repo/$O/prof_steps.log 2>&1; echo "prof steps rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_roofline -o r -- python /root/repo/bench.py --no-cpu-baseline --only roofline > /root/repo/$O/prof_roofline.log 2>&1; echo "prof roofline rc=$?"
cd /root/repo
python tools/trace_step.py $(find $O/prof_steps -name '*kernel_trace.csv' | head -1) 20 --timeline > $O/step_timeline.txt 2>&1
timeout 600 python bench.py --force-dist --steps 100 --warmup 10 > $O/bench_forcedist.json 2> $O/bench_forcedist.err; echo "forcedist rc=$?"; cat $O/bench_forcedist.json
timeout 900 python bench.py --workload synth --steps 10 --warmup 2 > $O/bench_synth_w1.json 2> $O/bench_synth_w1.err; echo "synth rc=$?"; cat $O/bench_synth_w1.json
timeout 600 python bench.py --workload tiktok --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_tiktok.json 2> $O/bench_tiktok.err; echo "tiktok rc=$?"
timeout 300 python tools/kbench.py --shape baby > $O/kbench.log 2>&1; echo "kbench rc=$?"; cp gpurun_out/kbench_baby_d64.json $O/ 2>/dev/null
timeout 600 python tools/trainer_bench.py --workload baby --batches 10 > $O/trainer_bench.log 2>&1; echo "trainer rc=$?"; grep -v amdgpu $O/trainer_bench.log | tail -1
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|FAILED" $O/pytest.log | head
find $O -name '*_kernel_trace.csv' -size +30M -delete
