for cfg in "ABC CAB 0,0,0" "ABC ABC 0,0,0" "CAB ABC 0,0,0" "CAB CAB 0,0,0" "ABC ABC 0,0,-1" "CAB ABC 0,0,-1" "ABC ABC -1,0,0"; do
  set -- $cfg
  r=$(MMSSL_FWD_ORDER=$1 MMSSL_BWD_ORDER=$2 MMSSL_STREAM_PRIO=$3 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
  echo "fwd=$1 bwd=$2 prio=$3 $r"
done
