#!/bin/bash
# round 5, GPU call 3: step timelines split vs f32
O=gpurun_out/r05c; mkdir -p $O; R=$(pwd); export TMPDIR=/tmp
for P in split f32; do
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/steps_$P -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --only steps --proj $P > /dev/null 2>&1
  cd $R
  python tools/trace_step.py $(find $O/steps_$P -name "*kernel_trace.csv" | head -1) 12 --timeline > $O/step_timeline_$P.txt 2>&1
  find $O/steps_$P -name "*kernel_stats.csv" -exec cp {} $O/steps_kernel_stats_$P.csv \;
  rm -rf $O/steps_$P
done
cat $O/step_timeline_split.txt
