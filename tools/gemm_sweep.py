"""Split-K sweep of the projection GEMM (run each setting in a fresh process: the env var is read once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, torch
sys.path.insert(0, %r)
from mmssl_amd import ops, _lib
M, d = 18357, 64
for K in (4096, 1024):
    F_ = torch.randn(M, K, device="cuda"); W = torch.randn(d, K, device="cuda") * 0.02; b = torch.zeros(d, device="cuda")
    keep = (torch.rand(M, d, device="cuda") >= 0.2).to(torch.uint8)
    gY = torch.randn(M, d, device="cuda"); gW = torch.empty_like(W); gb = torch.empty(d, device="cuda")
    nb = _lib.lib().mmssl_linear_wgrad_workspace_bytes(M, K, d); ws = torch.empty(nb // 4, device="cuda")
    def fwd(): ops.linear(F_, W, b, keep, 1.25)
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
    print("K=%%d fwd %%.1f us (%%.0f TF)  wgrad %%.1f us (%%.0f TF)" %% (K, res[0], fl / res[0] * 1e-6, res[1], fl / res[1] * 1e-6))
''' % ROOT
for sp in sys.argv[1:] or ["0", "2", "3", "5", "7", "8", "14"]:
    env = dict(os.environ)
    if sp != "0":
        env["MMSSL_GEMM_SPLITS"] = sp
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print("splits=%s (0=auto):" % sp, " | ".join(l for l in r.stdout.splitlines() if l.startswith("K=")), r.stderr[-300:] if r.returncode else "")
