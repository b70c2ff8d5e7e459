#!/bin/bash
O=gpurun_out/r05o; mkdir -p $O
python tools/projx_ablate.py 2>/dev/null | tee $O/ablate.txt
for d in 1 2 4 6 8 9 15; do python tools/projx_ablate.py tools/_dbg/libmmssl_xdbg$d.so 2>/dev/null | tee -a $O/ablate.txt; done
