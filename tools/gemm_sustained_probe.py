"""Steady-state (2 s sustained loops) timing of the projection forward / weight gradient for one kernel generation
(MMSSL_GEMM_V, MMSSL_GEMM_PP_BK) - short bursts of 100 launches measure 20-25 % slower than the sustained rate."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops


def run(fn, secs=float(os.environ.get("PROBE_SECS", "1.5"))):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.time()
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n



def main():
    tag = "V=%s BK=%s" % (os.environ.get("MMSSL_GEMM_V", "6"), os.environ.get("MMSSL_GEMM_PP_BK", "-"))
    gen = torch.Generator(device="cuda").manual_seed(0)
    out = []
    with torch.no_grad():
        for name, M, K in (("img", 18357, 4096), ("txt", 18357, 1024)):
            F_ = torch.randn(M, K, device="cuda", generator=gen)
            W = torch.randn(64, K, device="cuda", generator=gen) * 0.02
            b = torch.zeros(64, device="cuda")
            keep = (torch.rand(M, 64, device="cuda", generator=gen) >= 0.2).to(torch.uint8)
            gY = torch.randn(M, 64, device="cuda", generator=gen)
            fwd = run(lambda: ops._linear_raw(F_, W, b, keep, 1.25))
            if os.environ.get("PROBE_FT_FWD") == "1":
                os.environ["MMSSL_FWD_FT"] = "1"
                ops.register_transposed_features(F_)
                y1 = ops._linear_raw(F_, W, b, keep, 1.25)
                os.environ["MMSSL_FWD_FT"] = "0"
                y0 = ops._linear_raw(F_, W, b, keep, 1.25)
                os.environ["MMSSL_FWD_FT"] = "1"
                err = float((y1 - y0).norm() / y0.norm())
                fwd_ft = run(lambda: ops._linear_raw(F_, W, b, keep, 1.25))
                os.environ["MMSSL_FWD_FT"] = "0"
                ops._FT.clear()
                print("   %s fwd via F^T %.1f us (rel. diff to default %.2e)" % (name, fwd_ft, err), flush=True)
            os.environ["MMSSL_WGRAD_FT"] = "0"
            wg = run(lambda: ops._linear_wgrad_raw(gY, keep, 1.25, F_, W))
            wf = float("nan")
            if os.environ.get("MMSSL_GEMM_V", "6") != "5":
                os.environ["MMSSL_WGRAD_FT"] = "1"
                ops.register_transposed_features(F_)
                wf = run(lambda: ops._linear_wgrad_raw(gY, keep, 1.25, F_, W))
                ops._FT.clear()
            out.append("%s fwd %.1f us (%.0f TF) | wgrad reg-staged %.1f us | wgrad via F^T %.1f us" % (
                name, fwd, 2.0 * M * K * 64 / fwd * 1e-6, wg, wf))
            del F_
    print(tag, " || ".join(out), flush=True)


if __name__ == "__main__":
    main()
