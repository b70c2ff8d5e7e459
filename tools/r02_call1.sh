#!/bin/bash
# round-2 GPU call 1: evidence that needs no new code (clock/power probe of the GEMM, SQ stall counters,
# the configs[4] per-rank SpMM shape incl. FETCH/WRITE passes, kbench re-run under a kernel trace)
export TMPDIR=/tmp
O=gpurun_out/r02c1; mkdir -p $O
python tools/gemm_clock_probe.py > $O/gemm_clock_probe.txt 2>&1
echo "clock probe rc=$?"
PMC1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
PMC2="SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"
PMC3="GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$PMC1" "$PMC2" "$PMC3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/gemm_pmc$i -o g -- python tools/gemm_pmc.py > $O/gemm_pmc$i.log 2>&1
  echo "gemm pmc$i rc=$?"
  python tools/pmc_table.py $(find $O/gemm_pmc$i -name '*counter_collection.csv' | head -1) gemm > $O/gemm_pmc${i}_table.txt 2>&1
done
timeout 900 python tools/spmm_rank_shape.py > $O/spmm_rank_shape.log 2>&1
echo "rank shape rc=$?"; cp gpurun_out/spmm_rank_shape.json $O/ 2>/dev/null
for C in FETCH_SIZE WRITE_SIZE; do
  MODE=pmc timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/rank_$C -o r -- python tools/spmm_rank_shape.py > $O/rank_$C.log 2>&1
  echo "rank $C rc=$?"
  python tools/pmc_table.py $(find $O/rank_$C -name '*counter_collection.csv' | head -1) spmm > $O/rank_${C}_table.txt 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kbench_trace -o k -- python tools/kbench.py --shape baby > $O/kbench.log 2>&1
echo "kbench rc=$?"
find $O -name '*_kernel_trace.csv' -size +20M -delete
find $O -name '*counter_collection.csv' -size +8M -delete
du -sh $O
