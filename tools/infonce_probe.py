"""The hot step's InfoNCE loss chain alone (two problems sharing z2, n = batch rows gathered from the tables): sustained
time of each phase - forward row terms (prep + pair tiles + row terms), backward pair tiles, backward finish - and a
check of losses and gradients against a float64 torch restatement of batched_contrastive_loss
(/root/reference/MMSSL/main.py:222-245 with the 1e-8 inside the logarithm). MMSSL_LIB picks an A/B build."""
import argparse
import ctypes as ct
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--d", type=int, default=64)
ap.add_argument("--rows", type=int, default=35598)
ap.add_argument("--tau", type=float, default=0.5)
ap.add_argument("--secs", type=float, default=0.5)
a = ap.parse_args()
torch.manual_seed(0)
dev = "cuda"
L = _lib.lib()
P, n, d = 2, a.n, a.d
z1 = [torch.randn(a.rows, d, device=dev) for _ in range(P)]
z2 = torch.randn(a.rows, d, device=dev)
idx = torch.randperm(a.rows, device=dev)[:n].contiguous()
nb = L.mmssl_infonce_multi_workspace_bytes(P, n, d)
ws = torch.empty(nb // 4, device=dev)
losses = torch.zeros(P, device=dev)
gloss = torch.tensor([0.7, 1.3], device=dev)
g1 = [torch.zeros_like(z) for z in z1]
g2 = torch.zeros_like(z2)
z1s = (ct.c_void_p * P)(*[z.data_ptr() for z in z1])
g1s = (ct.c_void_p * P)(*[g.data_ptr() for g in g1])
sp = _lib.stream_ptr()
tickets = torch.zeros(8, dtype=torch.int32, device=dev)


def fwd():
    _lib.check(L.mmssl_infonce_multi_fwd_phase_f32(z1s, z2.data_ptr(), idx.data_ptr(), P, n, d, a.tau, losses.data_ptr(),
                                                   ws.data_ptr(), nb, 3, sp), "fwd")


def fwd_rows():        # the hot chain's form: the loss comes out of the last block to arrive (no reduction launch)
    _lib.check(L.mmssl_infonce_multi_fwd_ticket_f32(z1s, z2.data_ptr(), idx.data_ptr(), P, n, d, a.tau, losses.data_ptr(),
                                                    ws.data_ptr(), nb, tickets.data_ptr(), sp), "fwd_ticket")


def bwd(ph):
    _lib.check(L.mmssl_infonce_multi_bwd_phase_f32(idx.data_ptr(), P, n, d, a.tau, gloss.data_ptr(), g1s, g2.data_ptr(),
                                                   ws.data_ptr(), nb, ph, sp), "bwd%d" % ph)


def sustained(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    k, t0 = 0, time.time()
    e0.record()
    while time.time() - t0 < a.secs:
        for _ in range(50):
            fn()
        k += 50
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / k


# ---- check ----
fwd()
l_plain = losses.clone()
losses.zero_()
fwd_rows()
bwd(3)
torch.cuda.synchronize()
print("ticketed forward vs phased forward: loss difference %.2e" % (losses - l_plain).abs().max().item())
ref_l, ref_g1, ref_g2 = [], [], torch.zeros_like(z2, dtype=torch.float64)
for p in range(P):
    x = z1[p][idx].double().requires_grad_(True)
    y = z2[idx].double().requires_grad_(True)
    a1 = torch.nn.functional.normalize(x, dim=1)
    a2 = torch.nn.functional.normalize(y, dim=1)
    f = lambda s: torch.exp(s / a.tau)  # noqa: E731
    refl = f(a1 @ a1.t())
    btw = f(a1 @ a2.t())
    lo = -torch.log(btw.diag() / (refl.sum(1) + btw.sum(1) - refl.diag()) + 1e-8)
    loss = lo.mean()
    gx, gy = torch.autograd.grad(loss * gloss[p].double(), (x, y))
    ref_l.append(loss.item())
    ref_g1.append(gx)
    ref_g2.index_add_(0, idx, gy)
for p in range(P):
    el = abs(losses[p].item() - ref_l[p]) / abs(ref_l[p])
    eg = (g1[p][idx].double() - ref_g1[p]).abs().max().item() / ref_g1[p].abs().max().item()
    print("problem %d: loss rel err %.2e, g1 rel err %.2e" % (p, el, eg))
    assert el < 1e-5 and eg < 1e-4
eg2 = (g2.double() - ref_g2).abs().max().item() / ref_g2.abs().max().item()
print("g2 rel err %.2e" % eg2)
assert eg2 < 1e-4
print("n=%d d=%d: forward rows %.1f us, backward tiles %.1f us, backward finish %.1f us, whole chain %.1f us" % (
    n, d, sustained(fwd_rows), sustained(lambda: bwd(1)), sustained(lambda: bwd(2)),
    sustained(lambda: (fwd_rows(), bwd(3)))))
