"""Reproducer for the round-5 intermittent weight gradient (csrc/linear.hip, wgrad10_kernel), as an A/B of two builds:

    MMSSL_LIB=tools/_dbg/libmmssl_hip_r05.so python tools/wgrad_race_repro.py     # the round-5 kernel
    python tools/wgrad_race_repro.py                                               # the current one

The round-5 kernel kept eight steps of operand loads in flight in registers behind hand-counted `s_waitcnt vmcnt`; the
compiler copied one of those registers at the loop back-edge BEFORE the wait (tools/vmcnt_check.py shows the `v_mov_b64`).
The copy is only wrong when the load is late, i.e. when the memory system is saturated by OTHER kernels - so this tool
loops the per-modality weight gradient of configs[4] ([1M, 128]^T x [1M, 128]) on one stream while `hogs` other streams
run HBM-resident transposed SpMMs (the configs[4] rank operand, 512 MB gathered table) and large device copies, and
compares every result bit for bit with the one computed on an idle device."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mmssl_amd import _lib, ops, synth  # noqa: E402
from mmssl_amd.graph import GraphPlan  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=40)
    ap.add_argument("--hogs", type=int, default=3)
    ap.add_argument("--rows", type=int, default=1_000_000)
    a = ap.parse_args()
    dev = torch.device("cuda")
    raw = synth.interaction_matrix(250_000, 1_000_000, 12_500_000, seed=1000, item_seed=77)
    plan = GraphPlan(synth.normalised_rows(raw), xcd_bands=-1)
    g = torch.Generator().manual_seed(0)
    G = torch.randn(a.rows, 128, generator=g).to(dev)
    F_ = torch.randn(a.rows, 128, generator=g).to(dev)
    keep = (torch.rand(a.rows, 128, generator=g) >= 0.2).to(torch.uint8).to(dev)
    W = torch.empty(128, 128, device=dev)
    Gu = [torch.randn(250_000, 128, generator=g).to(dev) for _ in range(a.hogs)]
    big = [torch.empty(256 << 20, dtype=torch.uint8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    ref = {"plain": ops._linear_wgrad_raw(G, None, 1.0, F_, W)[1].clone(),
           "masked": ops._linear_wgrad_raw(G, keep, 1.25, F_, W)[1].clone()}
    torch.cuda.synchronize()
    main_s = torch.cuda.Stream()
    hog_s = [torch.cuda.Stream() for _ in range(a.hogs)]
    copy_s = torch.cuda.Stream()
    bad = {"plain": 0, "masked": 0}
    worst = 0.0
    for it in range(a.rounds):
        for h, st in enumerate(hog_s):
            with torch.cuda.stream(st):
                for _ in range(6):
                    ops._spmm_raw(plan.twin(h + 1), True, Gu[h], ops.EPI_NONE)
        with torch.cuda.stream(copy_s):
            for _ in range(8):
                big[1].copy_(big[0])
        with torch.cuda.stream(main_s):
            outs = {"plain": [ops._linear_wgrad_raw(G, None, 1.0, F_, W)[1] for _ in range(3)],
                    "masked": [ops._linear_wgrad_raw(G, keep, 1.25, F_, W)[1] for _ in range(3)]}
        torch.cuda.synchronize()
        for k, lst in outs.items():
            for o in lst:
                if not torch.equal(o, ref[k]):
                    bad[k] += 1
                    d = float((o - ref[k]).abs().max()) / float(ref[k].abs().max())
                    worst = max(worst, d)
                    print("round %d %s: %.3e of the largest entry" % (it, k, d), flush=True)
    print(json.dumps({"lib": os.path.basename(_lib.LIB_PATH), "rounds": a.rounds, "launches_per_kind": 3 * a.rounds,
                      "hog_streams": a.hogs, "not_bit_equal_to_idle_result": bad, "worst_rel": worst}))


if __name__ == "__main__":
    main()
