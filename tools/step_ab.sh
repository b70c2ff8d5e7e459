#!/bin/bash
# usage: tools/step_ab.sh OUTFILE ROUNDS "ENV_A" "ENV_B" ...   -- alternating single-GPU bench runs (steps only), ms/step
out=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do
  for cfg in "$@"; do
    env $cfg timeout 300 python bench.py --no-cpu-baseline --only steps --steps 1000 --warmup 300 2>&1 | grep -v amdgpu | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print('[$cfg]', r['ms_per_step'])" | tee -a $out
  done
done
python - $out <<'PY'
import sys, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    if l.startswith("["):
        k, v = l.rsplit("]", 1)
        d[k + "]"].append(float(v))
for k, v in d.items():
    v = sorted(v)
    print("SUMMARY %-50s min %.4f median %.4f max %.4f (n=%d)" % (k, v[0], v[len(v) // 2], v[-1], len(v)))
PY
