"""Device time of the step-level kernels (AdamW over the Baby parameter set, dropout masks)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops  # noqa: E402
from mmssl_amd.optim import FusedAdamW  # noqa: E402


def timed(fn, n=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * n)


shapes = [(35598, 64), (18357, 64), (64, 4096), (64,), (64, 1024), (64,), (256, 64)]
for name, make in (("FusedAdamW", lambda ps: FusedAdamW(ps, lr=5.5e-4)),
                   ("torch fused capturable", lambda ps: torch.optim.AdamW(ps, lr=5.5e-4, fused=True, capturable=True))):
    ps = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    for p in ps:
        p.grad = torch.randn_like(p)
    opt = make(ps)
    print("%-24s %.1f us/step" % (name, timed(opt.step)))
print("dropout_masks 2x[18357,64]  %.1f us" % timed(lambda: ops.dropout_masks(2, 18357, 64, 0.2, "cuda")))
print("torch bernoulli_ x2         %.1f us" % timed(lambda: [torch.empty((18357, 64), dtype=torch.uint8, device="cuda").bernoulli_(0.8) for _ in range(2)]))
