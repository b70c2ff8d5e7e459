"""us per call of the split-precision projection (forward and weight gradient, launch + epilogues) with one library:
python tools/projx_ablate.py [path/to/lib.so]  - used with the decomposition builds of tools/projx_ablate.sh."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mmssl_amd._lib as L
if len(sys.argv) > 1:
    L.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from mmssl_amd import ops
M, Ks = 18357, (4096, 1024)
torch.manual_seed(0)
Fs = [torch.randn(M, k, device="cuda") for k in Ks]
Ws = [torch.randn(64, k, device="cuda") * 0.02 for k in Ks]
G = torch.randn(M, 128, device="cuda")


def t(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) * 1e3 / n, 1)


print(json.dumps({"lib": os.path.basename(L.LIB_PATH), "forward_us": t(lambda: ops.proj_forward(Fs, Ws, [None, None])),
                  "wgrad_us": t(lambda: ops.proj_wgrad(G, Fs))}))
