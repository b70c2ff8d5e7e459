"""Sustained timing of the projection kernels on the Baby shapes: the per-modality kernels (stream-K forward + reduce,
register-direct weight gradient + reduce; image then text, back to back) against the grouped launch of
csrc/projection.hip. HIP events over `--secs`-long loops on one stream (bursts read 20-25 % slower: DESIGN.md)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--secs", type=float, default=1.0)
ap.add_argument("--M", type=int, default=18357)
ap.add_argument("--K", default="4096,1024")
ap.add_argument("--only-new", action="store_true")
a = ap.parse_args()
Ks = [int(x) for x in a.K.split(",")]
M = a.M
torch.manual_seed(0)
Fs = [torch.randn(M, k, device="cuda") for k in Ks]
Ws = [torch.randn(64, k, device="cuda") * 0.02 for k in Ks]
bs = [torch.zeros(64, device="cuda") for _ in Ks]
keep = (torch.rand(len(Ks), M, 64, device="cuda") >= 0.2).to(torch.uint8)
G = torch.randn(M, 64 * len(Ks), device="cuda") * keep.permute(1, 0, 2).reshape(M, -1).float()
Gs = [G[:, 64 * g:64 * g + 64].contiguous() for g in range(len(Ks))]
flops = sum(2.0 * M * k * 64 for k in Ks)


def sustained(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n, t0 = 0, time.time()
    e0.record()
    while time.time() - t0 < a.secs:
        for _ in range(20):
            fn()
        n += 20
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def old_fwd():
    for g in range(len(Ks)):
        ops._linear_raw(Fs[g], Ws[g], bs[g], keep[g], 1.25)


def new_fwd():
    ops.proj_forward(Fs, Ws, bs, keep=keep, scale=1.25)


def old_wgrad():
    for g in range(len(Ks)):
        ops._linear_wgrad_raw(Gs[g], keep[g], 1.25, Fs[g], Ws[g])


def new_wgrad():
    ops.proj_wgrad(G, Fs)


out = {"M": M, "K": Ks, "GFLOP": flops * 1e-9}
with torch.no_grad():
    cases = (("old_fwd_per_modality", old_fwd), ("grouped_fwd", new_fwd), ("old_wgrad_per_modality", old_wgrad),
             ("grouped_wgrad", new_wgrad))
    for name, fn in cases:
        if a.only_new and name.startswith("old"):
            continue
        us = sustained(fn)
        out[name] = {"us": round(us, 1), "TF": round(flops / us * 1e-6, 1), "frac_of_157.3TF": round(flops / us * 1e-6 / 157.3, 3)}
print(json.dumps(out))
