"""GEMM-only driver for rocprofv3 --pmc passes: img-shaped forward projection + wgrad, 5 launches each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops, _lib
M, K, d = 18357, int(os.environ.get("K", "4096")), 64
F_ = torch.randn(M, K, device="cuda"); W = torch.randn(d, K, device="cuda") * 0.02; b = torch.zeros(d, device="cuda")
gY = torch.randn(M, d, device="cuda"); gW = torch.empty_like(W); gb = torch.empty(d, device="cuda")
nb = _lib.lib().mmssl_linear_wgrad_workspace_bytes(M, K, d); ws = torch.empty(nb // 4, device="cuda")
for _ in range(5):
    ops.linear(F_, W, b)
    _lib.lib().mmssl_linear_wgrad_f32(gY.data_ptr(), None, 1.0, F_.data_ptr(), M, K, d, gW.data_ptr(), gb.data_ptr(), ws.data_ptr(), nb, _lib.stream_ptr())
torch.cuda.synchronize()
