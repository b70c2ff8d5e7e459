#!/bin/bash
# One round's evidence in ONE gpurun call (outputs under gpurun_out/<ROUND>ev; copy what matters to profiles/ by hand):
#   bash tools/evidence.sh ROUND [tests] [bench] [dist] [prof] [pmc] [hbm]      e.g.  bash tools/evidence.sh r04 tests bench
#     tests : the GPU test suite + smoke()
#     bench : the driver's own command, then the default line (1000 steps after 200 warm-up steps)
#     dist  : the sharded code path on one rank: item-side / gather-both, forced RCCL launches, column chunks; configs[4]'s
#             per-rank share as a whole step
#     prof  : rocprofv3 --kernel-trace --stats of the driver command (steps / roofline separately) + one step's timeline
#     pmc   : PMC passes (counters only with --kernel-trace): SpMM traffic (d = 64 / 128), projection MFMA / stalls / traffic
#     hbm   : the HBM-resident SpMM (configs[4] rank shape): FETCH_SIZE / WRITE_SIZE passes -> <ROUND>_spmm_hbm_pmc.json
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; export TMPDIR=/tmp
ROUND=${1:-r04}; shift
WHAT=" ${*:-tests bench dist prof} "
O=gpurun_out/${ROUND}ev; mkdir -p $O
has() { [[ "$WHAT" == *" $1 "* ]]; }
line() { python - "$1" <<'PY'
import json, sys
try:
    b = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][0])
    keys = ("ms_per_step", "value")
    print(sys.argv[1], {k: b.get(k) for k in keys}, "roofline", (b.get("roofline") or {}).get("frac"), "gcn", (b.get("gcn_forward") or {}).get("frac_hbm"),
          "proj", (b.get("projection") or {}).get("forward"), (b.get("projection") or {}).get("weight_gradient"),
          "loss", (b.get("loss_check") or {}).get("rel_err"), "hbm", {k: v.get("us") for k, v in (b.get("spmm_hbm") or {}).items() if isinstance(v, dict)},
          "comm", {k: (b.get("comm") or {}).get(k) for k in ("scheme", "column_chunks", "collectives_per_step", "bytes_per_step", "comm_only_ms")})
except Exception as e:
    print(sys.argv[1], "UNREADABLE", e)
PY
}
if has tests; then
  MMSSL_TEST_KEEP=$R/$O/nccl timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $O/gpu_tests.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O/gpu_tests.log
fi
if has bench; then
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err; line $O/bench_driver_cmd.json
  timeout 600 python bench.py > $O/bench_n1.json 2>> $O/bench.err; line $O/bench_n1.json
fi
if has dist; then
  B="python bench.py --gpus 1 --steps 500 --warmup 100 --no-cpu-baseline --only steps"
  timeout 300 $B > $O/bench_steps.json 2>> $O/bench.err; line $O/bench_steps.json
  timeout 300 $B --force-dist > $O/bench_forcedist.json 2>> $O/bench.err; line $O/bench_forcedist.json
  MMSSL_DIST_FORCE_COLLECTIVES=1 timeout 300 $B --force-dist --transport peer > $O/bench_forcedist_peer.json 2>> $O/bench.err; line $O/bench_forcedist_peer.json
  MMSSL_DIST_FORCE_COLLECTIVES=1 timeout 300 $B --force-dist --scheme item-side > $O/bench_forcedist_rccl_itemside.json 2>> $O/bench.err; line $O/bench_forcedist_rccl_itemside.json
  MMSSL_DIST_FORCE_COLLECTIVES=1 timeout 300 $B --force-dist --scheme item-side --chunks 2 > $O/bench_forcedist_rccl_itemside_c2.json 2>> $O/bench.err; line $O/bench_forcedist_rccl_itemside_c2.json
  MMSSL_DIST_FORCE_COLLECTIVES=1 timeout 300 $B --force-dist --scheme gather-both > $O/bench_forcedist_rccl_gatherboth.json 2>> $O/bench.err; line $O/bench_forcedist_rccl_gatherboth.json
  S="python bench.py --gpus 1 --workload synth --steps 10 --warmup 2 --no-cpu-baseline"
  timeout 600 $S > $O/bench_synth_w1.json 2>> $O/bench.err; line $O/bench_synth_w1.json
  MMSSL_DIST_FORCE_COLLECTIVES=1 timeout 600 $S --scheme item-side --chunks 2 > $O/bench_synth_w1_rccl_itemside_c2.json 2>> $O/bench.err; line $O/bench_synth_w1_rccl_itemside_c2.json
  MMSSL_DIST_FORCE_COLLECTIVES=1 timeout 600 $S --scheme item-side --chunks 4 > $O/bench_synth_w1_rccl_itemside_c4.json 2>> $O/bench.err; line $O/bench_synth_w1_rccl_itemside_c4.json
  MMSSL_DIST_FORCE_COLLECTIVES=1 timeout 600 $S --scheme gather-both > $O/bench_synth_w1_rccl_gatherboth.json 2>> $O/bench.err; line $O/bench_synth_w1_rccl_gatherboth.json
  timeout 600 python bench.py --gpus 1 --steps 500 --warmup 100 --no-cpu-baseline --no-hbm --graph communities > $O/bench_communities.json 2>> $O/bench.err; line $O/bench_communities.json
  # configs[4] WHOLE on one GPU (2M x 1M x 100M edges, d = 128): the N = 1 denominator of the 8-rank job's speed-up
  timeout 900 python bench.py --workload synth-full --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_synth_full_n1.json 2>> $O/bench.err; line $O/bench_synth_full_n1.json
  timeout 300 $B --proj f32 > $O/bench_steps_proj_f32.json 2>> $O/bench.err; line $O/bench_steps_proj_f32.json
fi
if has prof; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/steps -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --only steps > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/roofline -o t -- python $R/bench.py --gpus 1 --no-cpu-baseline --no-hbm --only roofline > /dev/null 2>&1
  cd $R
  python tools/trace_step.py $(find $O/steps -name "*kernel_trace.csv" | head -1) 12 --timeline > $O/step_timeline.txt 2>&1
  find $O/steps -name "*kernel_stats.csv" -exec cp {} $O/${ROUND}_rocprofv3_steps_kernel_stats.csv \;
  find $O/roofline -name "*kernel_stats.csv" -exec cp {} $O/${ROUND}_rocprofv3_roofline_kernel_stats.csv \;
  head -25 $O/${ROUND}_rocprofv3_steps_kernel_stats.csv
fi
if has pmc; then
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/spmm_$c -o p -- python $R/tools/spmm_pmc.py > /dev/null 2>&1
    D=128 timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/spmm128_$c -o p -- python $R/tools/spmm_pmc.py > /dev/null 2>&1
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/proj_$c -o p -- python $R/tools/proj_pmc.py > /dev/null 2>&1
  done
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/$O/proj_sq -o p -- python $R/tools/proj_pmc.py > /dev/null 2>&1
  cd $R
  python tools/pmc_summary.py $(find $O/spmm_FETCH_SIZE -name "*counter_collection.csv") $(find $O/spmm_WRITE_SIZE -name "*counter_collection.csv") $O/${ROUND}_spmm_pmc.json
  python tools/pmc_summary.py $(find $O/spmm128_FETCH_SIZE -name "*counter_collection.csv") $(find $O/spmm128_WRITE_SIZE -name "*counter_collection.csv") $O/${ROUND}_spmm_pmc_d128.json
  {
    echo "# rocprofv3 --pmc passes over tools/proj_pmc.py: the grouped projection kernels of the hot step, averages over the launches after the first"
    for p in proj_sq proj_FETCH_SIZE proj_WRITE_SIZE; do
      echo "## pass $p"; python tools/pmc_split.py $(find $O/$p -name "*counter_collection.csv") 8 proj
    done
  } > $O/${ROUND}_proj_pmc.txt 2>&1
fi
if has hbm; then
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    MODE=pmc timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/hbm_$c -o p -- python $R/tools/spmm_hbm_pmc.py > /dev/null 2>&1
  done
  cd $R
  python tools/spmm_hbm_pmc.py summarise $(find $O/hbm_FETCH_SIZE -name "*counter_collection.csv") $(find $O/hbm_WRITE_SIZE -name "*counter_collection.csv") $O/${ROUND}_spmm_hbm_pmc.json
fi
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
ls $O
