#!/bin/bash
# SQ counters of the InfoNCE tile kernels (LDS tile images) over tools/infonce_probe.py: bash tools/infonce_pmc.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/inf_pmc
mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$O/sq -o p -- python $R/tools/infonce_probe.py --secs 0.02 > /dev/null 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/$O/sq2 -o p -- python $R/tools/infonce_probe.py --secs 0.02 > /dev/null 2>&1
cd $R
{
  echo "# rocprofv3 --pmc passes over tools/infonce_probe.py (n = 1024, d = 64, two problems): the InfoNCE tile kernels with LDS"
  echo "# tile images. Counter values are sums over the chip per dispatch, averaged over the dispatches of the run."
  for p in sq sq2; do
    echo "## pass $p"; python tools/pmc_split.py $(find $O/$p -name "*counter_collection.csv") 100000 tiles_lds
  done
} > $O/r03_infonce_pmc.txt 2>&1
rm -rf $O/sq $O/sq2
cat $O/r03_infonce_pmc.txt
