"""Decomposition runs of the v7 ping-pong projection kernel (MMSSL_GEMM_V=7): which side limits it?
MMSSL_GEMM_PP_MODE bits: 1 = no LDS-DMA, 2 = no MFMA, 4 = no fragment reads. Run under rocprofv3 --kernel-trace --stats
to separate gemm_pp_kernel from pp_reduce_kernel; HIP-event times (kernel + reduce) are printed too."""
import os
import sys

import torch

os.environ["MMSSL_GEMM_V"] = "7"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops  # noqa: E402

M, K, d = 18357, 4096, 64
F_ = torch.randn(M, K, device="cuda")
W = torch.randn(d, K, device="cuda") * 0.02
b = torch.zeros(d, device="cuda")


def timeit(fn, iters=40, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


names = {0: "everything", 1: "no DMA (MFMA + fragment reads + barriers)", 2: "no MFMA (DMA + fragment reads + barriers)",
         4: "no fragment reads (DMA + MFMA)", 6: "DMA + barriers only", 5: "MFMA + barriers only", 3: "fragment reads + barriers only",
         7: "barriers only"}
with torch.no_grad():
    for mode in (0, 1, 2, 4, 6, 5, 3, 7, 0):
        os.environ["MMSSL_GEMM_PP_MODE"] = str(mode)
        us = timeit(lambda: ops._linear_raw(F_, W, b, None, 1.0))
        print("mode %d  %-46s %7.1f us (kernel + reduce, back to back)" % (mode, names[mode], us), flush=True)
os.environ["MMSSL_GEMM_PP_MODE"] = "0"
