# usage: bash tools/run_ab.sh "ENV1=.. ENV2=.." "ENV.." ...   (each argument = one configuration; 2 rounds)
for round in 1 2; do
  for cfg in "$@"; do
    r=$(env $cfg python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep -o '"ms_per_step": [0-9.]*')
    echo "round $round [$cfg] $r"
  done
done
