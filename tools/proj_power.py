"""rocm-smi power / sclk while the grouped projection (csrc/projection.hip) runs back to back for ~3 s per direction.
Run under the default library and under decomposition builds (tools/proj_variants.sh, MMSSL_LIB) to see whether the
matrix pipe and the DMA stream share a power / clock envelope when they run together."""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops  # noqa: E402

M, Ks = 18357, (4096, 1024)
torch.manual_seed(0)
Fs = [torch.randn(M, k, device="cuda") for k in Ks]
Ws = [torch.randn(64, k, device="cuda") * 0.02 for k in Ks]
bs = [torch.zeros(64, device="cuda") for _ in Ks]
keep = (torch.rand(len(Ks), M, 64, device="cuda") >= 0.2).to(torch.uint8)
G = torch.randn(M, 128, device="cuda") * keep.permute(1, 0, 2).reshape(M, -1).float()


def smi():
    out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True).stdout
    try:
        d0 = list(json.loads(out).values())[0]
    except Exception:
        return {}
    keep_ = {}
    for k, v in d0.items():
        kl = k.lower()
        if "sclk" in kl or "power" in kl:
            keep_[k.split("(")[0].strip()[:24]] = v
    return keep_


def run(name, fn, secs=3.0):
    stop, count = [False], [0]

    def loop():
        with torch.no_grad():
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
    print(json.dumps({"what": name, "lib": os.environ.get("MMSSL_LIB", "default"), "us_per_call": round(dt / count[0] * 1e6, 1),
                      "samples": samples[-4:]}), flush=True)


run("proj_forward", lambda: ops.proj_forward(Fs, Ws, bs, keep=keep, scale=1.25))
run("proj_wgrad", lambda: ops.proj_wgrad(G, Fs))
