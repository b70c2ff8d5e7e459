#!/bin/bash
# The round's evidence in one gpurun call (outputs under gpurun_out/r03ev, copied to profiles/ by hand afterwards):
#   bash tools/r03_evidence.sh [quick]      quick = skip the full GPU test suite
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=gpurun_out/r03ev; mkdir -p $O
if [ "$1" != "quick" ]; then
  timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/gpu_tests.log
fi
# 1. the driver's own command, then the default line (1000 steps after 200 warm-up steps)
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err
timeout 400 python bench.py > $O/bench_n1.json 2>> $O/bench.err
python -c "
import json
for f in ('bench_driver_cmd','bench_n1'):
    b=json.load(open('$O/%s.json'%f)); print(f, b['ms_per_step'], b['value'], b['roofline']['frac'], b['gcn_forward']['frac_hbm'], b['projection']['forward'], b['projection']['weight_gradient'], b['loss_check']['rel_err'])"
timeout 300 python bench.py --gpus 1 --steps 500 --warmup 100 --no-cpu-baseline --only steps --force-dist > $O/bench_forcedist.json 2>> $O/bench.err
timeout 300 python bench.py --gpus 1 --steps 500 --warmup 100 --no-cpu-baseline --only steps --dense-fuse > $O/bench_dense_fuse.json 2>> $O/bench.err
# 2. rocprofv3 kernel stats of the same commands (steps / roofline separately) + one step's timeline
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/steps -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --only steps > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/roofline -o t -- python $R/bench.py --gpus 1 --no-cpu-baseline --only roofline > /dev/null 2>&1
cd $R
python tools/trace_step.py $O/steps/t_kernel_trace.csv 12 --timeline > $O/step_timeline.txt 2>&1
# 3. PMC passes (counters only with --kernel-trace): SpMM traffic, projection MFMA / stalls / traffic
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/spmm_$c -o p -- python $R/tools/spmm_pmc.py > /dev/null 2>&1
  D=128 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/spmm128_$c -o p -- python $R/tools/spmm_pmc.py > /dev/null 2>&1
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/proj_$c -o p -- python $R/tools/proj_pmc.py > /dev/null 2>&1
done
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$O/proj_sq -o p -- python $R/tools/proj_pmc.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum SQ_INST_LEVEL_VMEM SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/$O/proj_sq2 -o p -- python $R/tools/proj_pmc.py > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $(find $O/spmm_FETCH_SIZE -name "*counter_collection.csv") $(find $O/spmm_WRITE_SIZE -name "*counter_collection.csv") $O/r03_spmm_pmc.json
python tools/pmc_summary.py $(find $O/spmm128_FETCH_SIZE -name "*counter_collection.csv") $(find $O/spmm128_WRITE_SIZE -name "*counter_collection.csv") $O/r03_spmm_pmc_d128.json
{
  echo "# rocprofv3 --pmc passes over tools/proj_pmc.py: the grouped projection kernels of the hot step (both Baby modalities"
  echo "# in one stream-K launch + one epilogue launch each way), averages over the launches after the first"
  for p in proj_sq proj_sq2 proj_FETCH_SIZE proj_WRITE_SIZE; do
    echo "## pass $p"; python tools/pmc_split.py $(find $O/$p -name "*counter_collection.csv") 8 proj_
  done
} > $O/r03_proj_pmc.txt 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
ls $O
