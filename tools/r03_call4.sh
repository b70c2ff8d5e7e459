#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c4; mkdir -p $O
R=$GRAFT_REPO_ROOT
for w in 8 4; do
  echo "== waves $w full"; MMSSL_PROJ_WAVES=$w timeout 120 python tools/proj_probe.py --only-new --secs 0.5
  for d in 1 2 3 4 5 6; do echo "== waves $w dbg $d (1 no DMA, 2 no MFMA, 4 no frag reads)"; MMSSL_LIB=$R/tools/_dbg/libmmssl_dbg$d.so MMSSL_PROJ_WAVES=$w timeout 120 python tools/proj_probe.py --only-new --secs 0.5; done
done 2>&1 | grep -v amdgpu.ids | tee $O/ablate.txt
cd /tmp
for w in 8 4; do
MMSSL_PROJ_WAVES=$w timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_w$w -o g -- python $R/tools/proj_probe.py --only-new --secs 0.05 > /dev/null 2>&1
MMSSL_PROJ_WAVES=$w timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $R/$O/pmc2_w$w -o g -- python $R/tools/proj_probe.py --only-new --secs 0.05 > /dev/null 2>&1
done
cd $R
for f in $(find $O -name "*counter_collection.csv"); do echo "== $f"; python tools/pmc_table.py $f proj_sk; done > $O/pmc_tables.txt 2>&1
cat $O/pmc_tables.txt
find $O -name "*.csv" -size +1M -delete
