#!/bin/bash
O=gpurun_out/r05h; mkdir -p $O
timeout 600 python tools/spmm_band_probe.py 2> $O/band.err | tee $O/spmm_band_probe.json | head -c 3000; echo
timeout 900 python tools/trainer_bench.py --workload baby --batches 12 2> $O/trainer.err | tail -1 | tee $O/trainer_loop_baby.json | head -c 2500; echo
timeout 600 python bench.py --steps 200 --warmup 50 --no-hbm > $O/bench_frows.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r05h/bench_frows.json") if l.startswith("{")][0])
print(json.dumps(d.get("f_rows"), indent=1)); print(d["ms_per_step"], d["loss_check"])
PY
tail -3 $O/bench.err
