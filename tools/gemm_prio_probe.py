"""s_setprio experiment on the projection kernels: v6 (MMSSL_GEMM_PRIO) and v7 (MMSSL_GEMM_PP_MODE bit 3)."""
import os, sys, subprocess
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from mmssl_amd import ops
M, K, d = 18357, 4096, 64
F_ = torch.randn(M, K, device="cuda"); W = torch.randn(d, K, device="cuda") * 0.02; b = torch.zeros(d, device="cuda")
def t(fn, iters=60):
    for _ in range(8): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / iters
with torch.no_grad():
    print("%%s: %%.1f us" %% (sys.argv[1], t(lambda: ops._linear_raw(F_, W, b, None, 1.0))))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for tag, env in (("v6", {"MMSSL_GEMM_V": "6"}), ("v6+prio", {"MMSSL_GEMM_V": "6", "MMSSL_GEMM_PRIO": "1"}),
                 ("v7", {"MMSSL_GEMM_V": "7", "MMSSL_GEMM_PP_BK": "32"}), ("v7+prio", {"MMSSL_GEMM_V": "7", "MMSSL_GEMM_PP_BK": "32", "MMSSL_GEMM_PP_MODE": "8"}),
                 ("v7 mfma-only+prio", {"MMSSL_GEMM_V": "7", "MMSSL_GEMM_PP_BK": "32", "MMSSL_GEMM_PP_MODE": "13"}),
                 ("v7 no-dma+prio", {"MMSSL_GEMM_V": "7", "MMSSL_GEMM_PP_BK": "32", "MMSSL_GEMM_PP_MODE": "9"})):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD, tag], env=e, capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
