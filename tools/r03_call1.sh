#!/bin/bash
# round 3, GPU call 1: real-RCCL tests of the sharded path, baseline bench on this box, SQ counters of the DEFAULT
# projection kernels, FETCH_SIZE / WRITE_SIZE calibration on a gather of known bytes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c1; mkdir -p $O
MMSSL_TEST_KEEP=$O timeout 1500 python -m pytest tests/test_dist_nccl_gpu.py -x -q > $O/nccl_tests.log 2>&1; echo "nccl tests rc=$?" >> $O/nccl_tests.log
tail -5 $O/nccl_tests.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err; tail -c 600 $O/bench_driver_cmd.json
cd /tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $R/$O/pmc_gemm1 -o g -- python $R/tools/gemm_pmc.py > $R/$O/pmc_gemm1.log 2>&1
timeout 300 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/$O/pmc_gemm2 -o g -- python $R/tools/gemm_pmc.py > $R/$O/pmc_gemm2.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/calib_f -o c -- python $R/tools/fetch_calib.py > $R/$O/calib_known.json 2> $R/$O/calib_f.log
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/calib_w -o c -- python $R/tools/fetch_calib.py > /dev/null 2> $R/$O/calib_w.log
cd $R
for f in $(find $O -name "*counter_collection.csv"); do echo "== $f"; python tools/pmc_table.py $f | head -60; done > $O/pmc_tables.txt 2>&1
find $O -name "*.csv" -size +2M -delete
ls -la $O
