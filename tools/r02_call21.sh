#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r02c21; mkdir -p $O
for cfg in "MMSSL_GEMM_V=9" "MMSSL_GEMM_V=9 MMSSL_GEMM_R9_MODE=1" "MMSSL_GEMM_V=9 MMSSL_GEMM_R9_MODE=2" "MMSSL_GEMM_V=9 MMSSL_GEMM_R9_BLOCKS=512"; do
  env $cfg timeout 200 python tools/gemm_sustained_probe.py 2>&1 | grep -v amdgpu | tail -3 | sed "s/^/[$cfg] /" | tee -a $O/sustained.txt
done
timeout 120 python - <<'PY' 2>&1 | grep -v amdgpu | tee -a $O/sustained.txt
import sys, time, torch
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
from gemm_sustained_probe import run
F_ = torch.randn(18357, 4096, device="cuda")
out = torch.empty(4096, device="cuda")
print("torch.sum(F, dim=0) sustained %.1f us" % run(lambda: torch.sum(F_, dim=0, out=out)))
o2 = torch.empty(18357, device="cuda")
print("torch.sum(F, dim=1) sustained %.1f us" % run(lambda: torch.sum(F_, dim=1, out=o2)))
G = torch.empty_like(F_)
print("copy sustained %.1f us" % run(lambda: G.copy_(F_)))
PY
