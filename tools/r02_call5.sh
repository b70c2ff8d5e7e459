#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c5; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/mode_trace -o m -- python /root/repo/tools/gemm_mode_probe.py > /root/repo/$O/mode_probe.log 2>&1; echo "mode rc=$?"
cd /root/repo
grep "^mode" $O/mode_probe.log
python - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/r02c5/mode_trace/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "gemm_pp_kernel" in r["Kernel_Name"] or "pp_reduce" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# 9 modes x 45 launches each
pp=[ (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "gemm_pp" in r["Kernel_Name"]]
rd=[ (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows if "pp_reduce" in r["Kernel_Name"]]
n=45
for i,mode in enumerate((0,1,2,4,6,5,3,7,0)):
    a=pp[i*n+5:(i+1)*n]; b=rd[i*n+5:(i+1)*n]
    print("mode %d: gemm_pp_kernel avg %.1f us min %.1f | pp_reduce avg %.1f us" % (mode, sum(a)/len(a), min(a), sum(b)/len(b)))
PY
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py tests/test_bench_contract_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
find $O -name '*_kernel_trace.csv' -size +30M -delete
