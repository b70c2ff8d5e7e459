import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops
d, K = 64, 4096
for M in (2048, 4096, 8192, 16384, 18357, 36714, 73428):
    F_ = torch.randn(M, K, device="cuda"); W = torch.randn(d, K, device="cuda") * 0.02; b = torch.zeros(d, device="cuda")
    fn = lambda: ops.linear(F_, W, b)
    for _ in range(5): fn()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay(); e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10
    print("M=%6d F=%6.1f MB  %.1f us  %.1f TF  %.2f TB/s" % (M, M * K * 4 / 1e6, us, 2.0 * M * K * d / us * 1e-6, M * K * 4 / us * 1e-6))
    del F_
