"""Projection GEMM A/B on one MI355X: forward (mmssl_linear_f32) and weight gradient (register-staged kernel vs the
forward kernel against F^T) for the Baby / Tiktok projection shapes; correctness against an fp64 torch reference.
Run once per kernel generation:  MMSSL_GEMM_V=5 python tools/gemm_v6_probe.py ; MMSSL_GEMM_V=6 python tools/gemm_v6_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops, _lib  # noqa: E402

V = os.environ.get("MMSSL_GEMM_V", "6")
out = {"MMSSL_GEMM_V": V, "MMSSL_GEMM_SK_BLOCKS": os.environ.get("MMSSL_GEMM_SK_BLOCKS", "512")}


def timeit(fn, iters=100, warm=10):
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


def graph_time(fn, iters=50):
    """the same launches inside one hipGraph, replayed back to back (no host launch gaps)"""
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay()
        e0.record()
        for _ in range(iters // 10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters // 10 * 10)


gen = torch.Generator(device="cuda").manual_seed(1)
for name, M, K in (("baby_img", 18357, 4096), ("baby_txt", 18357, 1024), ("tiktok_txt", 6710, 768), ("tiktok_img", 6710, 128)):
    d = 64
    F_ = torch.randn(M, K, device="cuda", generator=gen)
    W = torch.randn(d, K, device="cuda", generator=gen) * 0.02
    b = torch.randn(d, device="cuda", generator=gen)
    keep = (torch.rand(M, d, device="cuda", generator=gen) >= 0.2).to(torch.uint8)
    gY = torch.randn(M, d, device="cuda", generator=gen)
    rec = {}
    with torch.no_grad():
        Y = ops._linear_raw(F_, W, b, keep, 1.25)
        ref = ((F_.double() @ W.double().t() + b.double()) * keep.double() * 1.25)
        rec["fwd_max_rel_err"] = float((Y.double() - ref).abs().max() / ref.abs().max())
        rec["fwd_us"] = round(timeit(lambda: ops._linear_raw(F_, W, b, keep, 1.25)), 1)
        rec["fwd_us_graph"] = round(graph_time(lambda: ops._linear_raw(F_, W, b, keep, 1.25)), 1)
        rec["fwd_TF"] = round(2.0 * M * K * d / rec["fwd_us_graph"] * 1e-6, 1)
        # weight gradient, register-staged kernel
        os.environ["MMSSL_WGRAD_FT"] = "0"
        _, gW0, gb0 = ops._linear_wgrad_raw(gY, keep, 1.25, F_, W)
        gm = gY.double() * keep.double() * 1.25
        refW, refb = gm.t() @ F_.double(), gm.sum(0)
        rec["wgrad_old_max_rel_err"] = float((gW0.double() - refW).abs().max() / refW.abs().max())
        rec["wgrad_old_us_graph"] = round(graph_time(lambda: ops._linear_wgrad_raw(gY, keep, 1.25, F_, W)), 1)
        # weight gradient through the forward kernel against F^T
        os.environ["MMSSL_WGRAD_FT"] = "1"
        ops.register_transposed_features(F_)
        _, gW1, gb1 = ops._linear_wgrad_raw(gY, keep, 1.25, F_, W)
        rec["wgrad_ft_max_rel_err"] = float((gW1.double() - refW).abs().max() / refW.abs().max())
        rec["gb_ft_max_rel_err"] = float((gb1.double() - refb).abs().max() / refb.abs().max())
        rec["wgrad_ft_us_graph"] = round(graph_time(lambda: ops._linear_wgrad_raw(gY, keep, 1.25, F_, W)), 1)
        rec["wgrad_ft_TF"] = round(2.0 * M * K * d / rec["wgrad_ft_us_graph"] * 1e-6, 1)
        ops._FT.clear()
    out[name] = rec
    print(name, rec, flush=True)
    del F_
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemm_v6_probe_V%s_B%s.json" % (V, out["MMSSL_GEMM_SK_BLOCKS"]), "w"), indent=1)
