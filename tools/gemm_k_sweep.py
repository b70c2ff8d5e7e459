import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops, _lib
d, M = 64, 18357
for K in (4096, 4128, 4160, 4224, 3968, 2048, 2080, 1024, 1056):
    F_ = torch.randn(M, K, device="cuda"); W = torch.randn(d, K, device="cuda") * 0.02; b = torch.zeros(d, device="cuda")
    gY = torch.randn(M, d, device="cuda"); gW = torch.empty_like(W); gb = torch.empty(d, device="cuda")
    nb = _lib.lib().mmssl_linear_wgrad_workspace_bytes(M, K, d); ws = torch.empty(nb // 4, device="cuda")
    def fwd(): ops.linear(F_, W, b)
    def wg(): _lib.lib().mmssl_linear_wgrad_f32(gY.data_ptr(), None, 1.0, F_.data_ptr(), M, K, d, gW.data_ptr(), gb.data_ptr(), ws.data_ptr(), nb, _lib.stream_ptr())
    res = []
    for fn in (fwd, wg):
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
        res.append(e0.elapsed_time(e1) * 10)
    fl = 2.0 * M * K * d
    print("K=%5d  fwd %6.1f us (%5.1f TF)   wgrad %6.1f us (%5.1f TF)" % (K, res[0], fl / res[0] * 1e-6, res[1], fl / res[1] * 1e-6))
    del F_
