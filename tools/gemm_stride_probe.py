"""Does the power-of-two row pitch of F (K = 4096 floats = 16 KB) cost bandwidth? Same row count, K = 3968 / 4096 /
4224 (pitch = K), sustained timing of the forward projection; time per 128 k-values is the comparable figure."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mmssl_amd import ops
from gemm_sustained_probe import run

gen = torch.Generator(device="cuda").manual_seed(0)
M = 18357
with torch.no_grad():
    for K in (3968, 4096, 4224, 8192, 8320, 1024, 1152):
        F_ = torch.randn(M, K, device="cuda", generator=gen)
        W = torch.randn(64, K, device="cuda", generator=gen) * 0.02
        b = torch.zeros(64, device="cuda")
        t = run(lambda: ops._linear_raw(F_, W, b, None, 1.0), secs=1.0)
        print("V=%s K=%5d pitch %6d B: fwd %.1f us = %.2f us per 128 k = %.0f TF" % (
            os.environ.get("MMSSL_GEMM_V", "6"), K, 4 * K, t, t / (K / 128), 2.0 * M * K * 64 / t * 1e-6), flush=True)
        del F_, W
