"""Opt-in split-precision projection (mmssl_linear_split_f32): accuracy vs an fp64 reference next to the exact
fp32 kernels, and time under a hipGraph. Forward Y = F.W^T and wgrad gW^T = F^T.gY (same kernel, transposed
operands padded along the reduction)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops, _lib  # noqa: E402

L = _lib.lib()


def split(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous()


def call_split(Ah, Al, Bh, Bl, M, K, N):
    Y = torch.empty(M, N, device="cuda")
    nb = L.mmssl_linear_split_workspace_bytes(M, K, N)
    ws = torch.empty(max(nb // 4, 4), device="cuda")
    def fn():
        rc = L.mmssl_linear_split_f32(Ah.data_ptr(), Al.data_ptr(), Bh.data_ptr(), Bl.data_ptr(), None, None, 1.0, M, K, N,
                                      Y.data_ptr(), ws.data_ptr(), nb, _lib.stream_ptr())
        _lib.check(rc, "mmssl_linear_split_f32")
    return fn, Y


def timed(fn, n=10):
    for _ in range(3):
        fn()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * n)


def rel(a, ref):
    return float((a.double() - ref).abs().max() / ref.abs().max())


M, d = 18357, 64
for K in (4096, 1024):
    torch.manual_seed(K)
    F_ = torch.randn(M, K, device="cuda")
    W = torch.randn(d, K, device="cuda") * 0.02
    ref = F_.double() @ W.double().t()
    Fh, Fl = split(F_)
    Wh, Wl = split(W)
    fn, Y = call_split(Fh, Fl, Wh, Wl, M, K, d)
    fn()
    torch.cuda.synchronize()
    y32 = ops.linear(F_, W)
    t_split = timed(fn)
    t_f32 = timed(lambda: ops.linear(F_, W))
    print("fwd K=%d: split %.1f us (err %.2e)   fp32 kernel %.1f us (err %.2e)" % (K, t_split, rel(Y, ref), t_f32, rel(y32, ref)))
    # wgrad through the same kernel: gW^T[K, d] = F^T[K, Mp] . gY^T[d, Mp]^T, reduction padded to a multiple of 128
    gY = torch.randn(M, d, device="cuda")
    Mp = (M + 127) // 128 * 128
    FT = torch.zeros(K, Mp, device="cuda"); FT[:, :M] = F_.t()
    gT = torch.zeros(d, Mp, device="cuda"); gT[:, :M] = gY.t()
    refw = gY.double().t() @ F_.double()
    FTh, FTl = split(FT)
    gTh, gTl = split(gT)
    fnw, gWd = call_split(gTh, gTl, FTh, FTl, d, Mp, K)
    fnw()
    torch.cuda.synchronize()
    nb = L.mmssl_linear_wgrad_workspace_bytes(M, K, d)
    wsw = torch.empty(nb // 4, device="cuda")
    gW = torch.empty(d, K, device="cuda"); gb = torch.empty(d, device="cuda")
    def w32():
        L.mmssl_linear_wgrad_f32(gY.data_ptr(), None, 1.0, F_.data_ptr(), M, K, d, gW.data_ptr(), gb.data_ptr(), wsw.data_ptr(), nb, _lib.stream_ptr())
    w32()
    torch.cuda.synchronize()
    print("wgrad K=%d: split %.1f us (err %.2e)   fp32 kernel %.1f us (err %.2e)" % (
        K, timed(fnw), rel(gWd, refw), timed(w32), rel(gW, refw)))
