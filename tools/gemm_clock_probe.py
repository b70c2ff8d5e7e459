"""Is the projection GEMM power-bound, clock-bound or memory-bound? Time it (and reference kernels) under
different sclk caps (rocm-smi --setperfdeterminism) and power caps (--setpoweroverdrive). A kernel whose time
scales with 1/sclk is issue/latency bound; one that only moves under a power cap is energy-bound; one that
does not move is memory-bound.   python tools/gemm_clock_probe.py > gpurun_out/gemm_clock_probe.txt"""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops, _lib  # noqa: E402

SMI = "/opt/rocm/bin/rocm-smi"
M, K, d = 18357, 4096, 64
F_ = torch.randn(M, K, device="cuda")
W = torch.randn(d, K, device="cuda") * 0.02
b = torch.zeros(d, device="cuda")
gY = torch.randn(M, d, device="cuda")
gW = torch.empty_like(W)
gb = torch.empty(d, device="cuda")
nb = _lib.lib().mmssl_linear_wgrad_workspace_bytes(M, K, d)
ws = torch.empty(nb // 4, device="cuda")
big = torch.randn(64 * 1024 * 1024, device="cuda")           # 256 MiB
big2 = torch.empty_like(big)
x = torch.randn(4096, 4096, device="cuda")


def smi_json(*a):
    out = subprocess.run([SMI, *a, "--json"], capture_output=True, text=True).stdout
    try:
        return list(json.loads(out).values())[0]
    except Exception:
        return {"raw": out[:300]}


def sample():
    d0 = smi_json("--showpower", "--showclocks")
    keep = {}
    for k, v in d0.items():
        kl = k.lower()
        if "sclk" in kl or "power" in kl or "mclk" in kl:
            keep[k.split("(")[0].strip()[:24]] = v
    return keep


def timed(fn, secs=1.5):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    samples = []
    stop = [False]

    def sampler():
        time.sleep(0.4)
        while not stop[0]:
            samples.append(sample())
            time.sleep(0.25)
    th = threading.Thread(target=sampler)
    th.start()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop[0] = True
    th.join()
    return e0.elapsed_time(e1) * 1e3 / n, samples[-2:] if samples else []


KERNELS = [
    ("gemm_fwd_img", lambda: ops.linear(F_, W, b)),
    ("gemm_wgrad_img", lambda: _lib.lib().mmssl_linear_wgrad_f32(gY.data_ptr(), None, 1.0, F_.data_ptr(), M, K, d,
                                                                 gW.data_ptr(), gb.data_ptr(), ws.data_ptr(), nb,
                                                                 _lib.stream_ptr())),
    ("hbm_copy_256MiB", lambda: big2.copy_(big)),
    ("rocblas_sgemm_4096", lambda: torch.mm(x, x)),
]


def setting(name, cmds):
    print("=== setting:", name, flush=True)
    for c in cmds:
        r = subprocess.run([SMI] + c, capture_output=True, text=True)
        tail = " | ".join(l for l in (r.stdout + r.stderr).splitlines() if l.strip() and "====" not in l)[-300:]
        print("   $ rocm-smi %s -> rc=%d %s" % (" ".join(c), r.returncode, tail), flush=True)
    time.sleep(0.5)
    for nm, fn in KERNELS:
        us, s = timed(fn)
        print("   %-20s %8.1f us   %s" % (nm, us, s), flush=True)


try:
    setting("default", [])
    setting("sclk cap 1900 MHz (perf determinism)", [["--setperfdeterminism", "1900"]])
    setting("sclk cap 1500 MHz (perf determinism)", [["--setperfdeterminism", "1500"]])
    setting("sclk cap 1100 MHz (perf determinism)", [["--setperfdeterminism", "1100"]])
    setting("reset determinism", [["--resetperfdeterminism"]])
    setting("power cap 1000 W", [["--setpoweroverdrive", "1000", "--autorespond", "y"]])
    setting("power cap 750 W", [["--setpoweroverdrive", "750", "--autorespond", "y"]])
finally:
    subprocess.run([SMI, "--resetpoweroverdrive"], capture_output=True)
    subprocess.run([SMI, "--resetperfdeterminism"], capture_output=True)
    subprocess.run([SMI, "-r"], capture_output=True)
setting("after reset", [])
