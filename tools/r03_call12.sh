#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r03c12; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 300 python tests/_nccl_worker.py g8 $O/g8.json > $O/g8.out 2> $O/g8.err; echo "g8 rc=$?"; head -c 3000 $O/g8.err; echo; python -c "
import json;print({k:v for k,v in json.load(open('$O/g8.json')).items() if 'error' in k or k=='ok'})" 2>/dev/null
timeout 600 python -m pytest tests/test_proj_gpu.py -x -q 2>&1 | tail -1
for v in 0 1; do
  echo "== wgrad 2sets=$v"
  MMSSL_PROJ_WGRAD_2SETS=$v timeout 120 python tools/proj_probe.py --only-new --secs 0.5 2>/dev/null
  for fa in "" "--no-fuse-adam"; do
    MMSSL_PROJ_WGRAD_2SETS=$v timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --only steps $fa 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('  $fa driver-cmd', b['ms_per_step'])"
    MMSSL_PROJ_WGRAD_2SETS=$v timeout 300 python bench.py --gpus 1 --steps 1000 --warmup 200 --no-cpu-baseline --only steps $fa 2>/dev/null | python -c "
import json,sys;b=json.loads(sys.stdin.read());print('  $fa long run', b['ms_per_step'])"
  done
done
