"""Projection-GEMM driver for rocprofv3 --pmc passes: the DEFAULT kernels of the step (stream-K forward with bias +
dropout epilogue, register-direct weight gradient with fused dropout backward + bias gradient) on the Baby image
([18357, 4096]) and text ([18357, 1024]) shapes, 6 launches each.
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ... --kernel-trace --output-format csv -d out -- python tools/gemm_pmc.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops  # noqa: E402

M, d = 18357, 64
torch.manual_seed(0)
for K in (4096, 1024):
    F_ = torch.randn(M, K, device="cuda")
    W = (torch.randn(d, K, device="cuda") * 0.02).requires_grad_(True)
    b = torch.zeros(d, device="cuda", requires_grad=True)
    keep = (torch.rand(M, d, device="cuda") >= 0.2).to(torch.uint8)
    gY = torch.randn(M, d, device="cuda")
    for _ in range(6):
        W.grad = b.grad = None
        y = ops.linear(F_, W, b, keep, 1.25)
        y.backward(gY)
    torch.cuda.synchronize()
    del F_
