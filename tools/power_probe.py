"""Clocks and socket power (rocm-smi) while one kernel family runs back to back for a few seconds:
is the projection GEMM limited by the chip's power management rather than by its own structure?
Workloads: hbm (streaming read+write of 1 GB), mfma (tools/mfma_peak binary if built), gemm_fwd, gemm_wgrad, spmm."""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops, _lib  # noqa: E402

M, K, d = 18357, 4096, 64
F_ = torch.randn(M, K, device="cuda")
W = torch.randn(d, K, device="cuda") * 0.02
b = torch.zeros(d, device="cuda")
gY = torch.randn(M, d, device="cuda")
gW = torch.empty_like(W)
gb = torch.empty(d, device="cuda")
nb = _lib.lib().mmssl_linear_wgrad_workspace_bytes(M, K, d)
ws = torch.empty(nb // 4, device="cuda")
big = torch.randn(256 * 1024 * 1024 // 4 * 4, device="cuda")           # 1 GiB
big2 = torch.empty_like(big)


def smi():
    out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True).stdout
    try:
        d0 = list(json.loads(out).values())[0]
    except Exception:
        return {"raw": out[:200]}
    keep = {}
    for k, v in d0.items():
        kl = k.lower()
        if any(t in kl for t in ("sclk", "mclk", "fclk", "socclk", "power")):
            keep[k.split("(")[0].strip()[:28]] = v
    return keep


def run(name, fn, secs=3.0):
    stop = [False]
    count = [0]

    def loop():
        while not stop[0]:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            count[0] += 20
    th = threading.Thread(target=loop)
    t0 = time.time()
    th.start()
    samples = []
    time.sleep(0.8)
    while time.time() - t0 < secs:
        samples.append(smi())
        time.sleep(0.3)
    stop[0] = True
    th.join()
    dt = time.time() - t0
    print("== %s: %.1f us per call" % (name, dt / max(count[0], 1) * 1e6))
    for s in samples[-3:]:
        print("   ", s)


print("idle", smi())
run("hbm copy 1 GiB", lambda: big2.copy_(big))
run("gemm fwd img", lambda: ops.linear(F_, W, b))
run("gemm wgrad img", lambda: _lib.lib().mmssl_linear_wgrad_f32(gY.data_ptr(), None, 1.0, F_.data_ptr(), M, K, d, gW.data_ptr(),
                                                             gb.data_ptr(), ws.data_ptr(), nb, _lib.stream_ptr()))
x = torch.randn(8192, 8192, device="cuda")
run("rocBLAS sgemm 8192^3", lambda: torch.mm(x, x))
xb = x.to(torch.bfloat16)
run("rocBLAS/hipBLASLt bf16 gemm 8192^3", lambda: torch.mm(xb, xb))
small = torch.randn(4 * 1024 * 1024, device="cuda")        # 16 MB: stays in L2 / Infinity Cache
small2 = torch.empty_like(small)
run("cache-resident copy 16 MB", lambda: small2.copy_(small))
