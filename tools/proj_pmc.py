"""Grouped-projection driver for rocprofv3 --pmc passes: the hot step's projection launches (csrc/projection.hip: one
stream-K launch + epilogue launch each way for BOTH Baby modalities, [18357, 4096] and [18357, 1024] -> 64 channels,
bias + dropout masks drawn in the epilogue / masked gradient + bias gradient), 8 launches each way.
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES ... --kernel-trace --output-format csv -d out -- python tools/proj_pmc.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops  # noqa: E402

M, Ks = 18357, (4096, 1024)
torch.manual_seed(0)
Fs = [torch.randn(M, k, device="cuda") for k in Ks]
Ws = [torch.randn(64, k, device="cuda") * 0.02 for k in Ks]
bs = [torch.zeros(64, device="cuda") for _ in Ks]
ops.seed_dropout(3)
st = ops._rng_state(torch.device("cuda"))
keep = (torch.rand(len(Ks), M, 64, device="cuda") >= 0.2).to(torch.uint8)
G = torch.randn(M, 64 * len(Ks), device="cuda") * keep.permute(1, 0, 2).reshape(M, -1).float()
with torch.no_grad():
    for _ in range(8):
        ops.proj_forward(Fs, Ws, bs, draw=(0.2, st), scale=1.25)
        ops.tick_rng(torch.device("cuda"))
    for _ in range(8):
        ops.proj_wgrad(G, Fs)
torch.cuda.synchronize()
