#!/usr/bin/env python
"""bench.py — the MMSSL hot path on N MI355X GPUs (one process per GPU).

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: MMSSL forward (V/T
projection, modal + 3-layer GCN SpMM propagation, fusion) -> BPR + 2x InfoNCE + feature
regulariser -> backward -> AdamW — the reference's generator step (main.py:363-429) without the
adjacent GAN pieces. Inputs (graph, features, parameters, batch indices) are resident in HBM
before the timed region. Workload at N=1: the Amazon-Baby shape BASELINE.json quotes the metric on
(35598 x 18357, 256308 edges, V4096/T1024, d=64, B=1024), seeded synthetic data. At N>1 the tables and CSR rows
are row-sharded with an RCCL all-gather before every propagation layer (reduce-scatter in the backward):
`--scaling weak` (default) = the shape x N, every rank generating only its own users' interactions;
`--scaling strong` = the Baby graph itself cut N ways (BASELINE configs[3]); `--workload synth` = the 100M-edge
d=128 stress shape of configs[4] (2M x 1M over 8 ranks: 250K users x 125K items x 12.5M edges per rank).

metric = edge.layers/s: d-wide multiply-adds per nonzero summed over EVERY SpMM launch of the step (forward and backward;
a launch over the packed 2d-wide modal features counts two per nonzero, like the two d-wide launches it replaces) / step
time, whole job. One JSON line is printed by rank 0. Besides the contract fields it carries
`roofline` (the CSR SpMM, isolated, HIP events), `gcn_forward` (the 6-SpMM 3-layer propagation alone under hipGraph:
the quantity north_star's ">= 60 % of the HBM roofline" target is stated on), `loss_check` (the first step's loss
with injected dropout masks against the CPU oracle), `cpu_baseline`, and at N>1 `comm`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X HBM3E spec (guides/MI355X_MICROARCH.md)


def make_batches(raw, n_batches, B, seed):
    """Synthetic BPR triples resident on the device: B distinct users, one of their items, one
    item they did not interact with (vectorised stand-in for Data.sample(); the sampler itself is
    host Python and outside the timed hot path)."""
    rng = np.random.default_rng(seed)
    U, I = raw.shape
    indptr, indices = raw.indptr, raw.indices
    active = np.nonzero(np.diff(indptr) > 0)[0]
    out = []
    for _ in range(n_batches):
        users = rng.choice(active, size=B, replace=B > active.shape[0])
        deg = indptr[users + 1] - indptr[users]
        pos = indices[indptr[users] + (rng.random(B) * deg).astype(np.int64)]
        neg = rng.integers(0, I, size=B)
        for _ in range(8):      # rejection against the user's own items
            bad = np.array([n in indices[indptr[u]:indptr[u + 1]] for u, n in zip(users, neg)])
            if not bad.any():
                break
            neg[bad] = rng.integers(0, I, size=int(bad.sum()))
        out.append((users.astype(np.int64), pos.astype(np.int64), neg.astype(np.int64)))
    return out


def build_single_gpu(a, dev):
    from mmssl_amd import config, synth
    from mmssl_amd.graph import GraphPlan
    from mmssl_amd.hotpath import HotPathStep
    from mmssl_amd.Models import MMSSL
    import scipy.sparse as sp
    U, I, E, dv, dt = synth.SHAPES[a.workload]
    config.configure([], embed_size=a.d, weight_size=str([a.d] * a.gcn_layers), batch_size=a.batch,
                     drop_rate=0.2, layers=1)
    if a.graph == "communities":     # the same shape with column locality (8 communities, 10 % global edges): XCD-banded plans
        raw = synth.interaction_matrix_communities(U, I, E, seed=1)
    else:
        raw = synth.interaction_matrix(U, I, E, seed=1)
    ui, iu = synth.normalised_pair(raw)
    plans = [GraphPlan(ui, xcd_bands=a.xcd_bands), GraphPlan(iu, xcd_bands=a.xcd_bands)]
    # steady state of the reference loop: modal graphs are empty from the third batch on
    e_ui = GraphPlan(sp.csr_matrix((U, I), dtype=np.float32))
    e_iu = GraphPlan(sp.csr_matrix((I, U), dtype=np.float32))
    graphs = (plans[0], plans[1], e_ui, e_iu, e_ui, e_iu)
    torch.manual_seed(2022)
    g = torch.Generator().manual_seed(7)
    img = torch.randn(I, dv, generator=g).numpy()
    txt = torch.randn(I, dt, generator=g).numpy()
    model = MMSSL(U, I, a.d, [a.d] * a.gcn_layers, [0.1] * a.gcn_layers, img, txt).to(dev)
    model.train()
    step = HotPathStep(model, graphs, a.batch, decay=1e-5, fuse_adam=not a.no_fuse_adam, batch_rows=not a.dense_fuse)
    return step, raw, (ui, iu), plans


def count_edge_layers(step, d):
    from mmssl_amd import ops
    ops.STATS.update(enabled=True, spmm_launches=0, edge_layers=0, spmm_bytes=0, unit_d=d)
    step.step()
    torch.cuda.synchronize()
    ops.STATS["enabled"] = False
    return dict(ops.STATS, edge_layers=int(round(ops.STATS["edge_layers"])))


def spmm_roofline(plans, mats, d, iters=200, traffic=True):
    # (the committed PMC pass behind `traffic` was taken on the uniform graph with the flat work list)
    """Average duration of the dominant kernel (the CSR SpMM, all four launch flavours of the
    step: A_ui, A_iu and their transposes) from HIP events on the launch stream, against the
    algorithmic bytes per launch (nnz*(8+4d) + rows*4d + (rows+1)*4)."""
    from mmssl_amd import ops, synth
    dev = "cuda"
    ui, iu = mats
    launches = [(plans[0], False, ui), (plans[1], False, iu), (plans[0], True, ui.T.tocsr()), (plans[1], True, iu.T.tocsr())]
    X = {r: torch.randn(r, d, device=dev) for r in {m.shape[1] for _, _, m in launches}}
    nbytes = [synth.spmm_bytes(m, d) for _, _, m in launches]
    with torch.no_grad():
        def one_round():
            for (p, t, m) in launches:
                ops.spmm(p, X[m.shape[1]], transpose=t)
        for _ in range(10):
            one_round()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                for _ in range(10):
                    one_round()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay()
        e0.record()
        for _ in range(iters // 10):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
    n_launch = (iters // 10) * 10 * len(launches)
    avg_us = e0.elapsed_time(e1) * 1e3 / n_launch
    avg_bytes = float(np.mean(nbytes))
    achieved = avg_bytes / avg_us * 1e-3      # GB/s
    rec = {"bound": "hbm", "kernel": "spmm_kernel<16> (CSR SpMM d=%d)" % d, "achieved": round(achieved, 1),
           "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
           "avg_launch_us": round(avg_us, 2), "algorithmic_bytes_per_launch": int(avg_bytes),
           "traffic": load_traffic() if traffic else None,
           "traffic_source": ("profiles/%s (rocprofv3 --pmc pass, FETCH_SIZE x 2 + WRITE_SIZE, gfx950-calibrated; taken by "
                              "tools/evidence.sh in its own run - counters cannot be collected beside the timed region - and "
                              "tied to the kernel sources by sha256)" % TRAFFIC_FILE[0]) if traffic and TRAFFIC_FILE[0] else None}
    if rec["frac"] > 1.0:
        rec["note"] = ("algorithmic bytes count every gathered row once per edge; here the gathered table fits the 256 MB "
                       "Infinity Cache, so most of those bytes never reach HBM and the ratio to the HBM peak exceeds 1")
    return rec


def gcn_forward_record(plans, mats, d, n_layers, iters=200):
    """The G-layer GCN propagation alone (Models.py:201-211: u = A_ui.i, i = A_iu.u, G times = 2G dependent
    SpMMs, softmax epilogue on the last layer) captured in one hipGraph and replayed back to back."""
    from mmssl_amd import ops, synth
    ui, iu = mats
    dev = "cuda"
    u0, i0 = torch.randn(ui.shape[0], d, device=dev), torch.randn(iu.shape[0], d, device=dev)

    def chain():
        u, i = u0, i0
        for l in range(n_layers):
            epi = ops.EPI_SOFTMAX if l == n_layers - 1 else ops.EPI_NONE
            u = ops.spmm(plans[0], i, epilogue=epi)
            i = ops.spmm(plans[1], u, epilogue=epi)
        return i
    with torch.no_grad():
        for _ in range(5):
            chain()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                chain()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            g.replay()
            e0.record()
            for _ in range(iters):
                g.replay()
            e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    nbytes = n_layers * (synth.spmm_bytes(ui, d) + synth.spmm_bytes(iu, d))
    el = n_layers * (ui.nnz + iu.nnz)
    gbps = nbytes / us * 1e-3
    return {"what": "%d-layer GCN forward = %d dependent SpMM, one hipGraph, replayed back to back" % (n_layers, 2 * n_layers),
            "us": round(us, 2), "edge_layers": int(el), "edge_layers_per_s": round(el / us * 1e6, 1),
            "algorithmic_bytes": int(nbytes), "achieved_GBps": round(gbps, 1), "frac_hbm": round(gbps / HBM_PEAK_GBPS, 4),
            "target_frac": 0.6}


def projection_record(step, iters=600):
    """The grouped modality projection alone (both modalities in one stream-K launch + epilogue): forward with bias +
    dropout drawn in the epilogue, weight gradient + bias gradient from a dropout-masked output gradient; HIP events
    around hipGraph replays, ~0.4 s per direction (a burst of a few hundred launches reads 10-20 % slower than the
    sustained rate: the clocks are still settling). fp32 MFMA peak 157.3 TFLOP/s (guides/MI355X_MICROARCH.md)."""
    from mmssl_amd import ops
    m = step.model
    Fs = [m.image_feats, m.text_feats]
    Ws = [m.image_trans.weight.detach(), m.text_trans.weight.detach()]
    bs = [m.image_trans.bias.detach(), m.text_trans.bias.detach()]
    M = Fs[0].shape[0]
    dev = Fs[0].device
    flops = sum(2.0 * M * f.shape[1] * 64 for f in Fs)
    nbytes = sum(4.0 * (M * f.shape[1] + 64 * f.shape[1] + M * 64) for f in Fs)       # F + W + Y, fp32
    split = bool(ops.PROJ_SPLIT and ops.projx_supported([f.shape[1] for f in Fs], M, 64))
    st = ops._rng_state(dev).clone()
    G = torch.randn(M, 128, device=dev) * (torch.rand(M, 128, device=dev) >= 0.2)      # like the step's masked gradient
    out = {"what": "grouped projection of both modalities, one stream-K launch + epilogue each way", "GFLOP": round(flops * 1e-9, 3),
           "peak_TFLOPs": 157.3, "algorithmic_MB": round(nbytes * 1e-6, 1),
           "arithmetic": ("split precision: every fp32 value cut exactly into 3 bf16 pieces, 6 partial products per product on "
                          "v_mfma_f32_32x32x16_bf16, fp32 accumulate (error vs float64 <= the fp32-MFMA kernels': "
                          "tests/test_proj_gpu.py); bound = the feature stream (HBM); frac_mfma = fp32-equivalent rate over the "
                          "fp32 MFMA peak, can exceed 1") if split else "fp32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32)",
           "bound": "hbm" if split else "mfma"}
    with torch.no_grad():
        for name, fn in (("forward", lambda: ops.proj_forward(Fs, Ws, bs, draw=(0.2, st), scale=1.25)),
                         ("weight_gradient", lambda: ops.proj_wgrad(G, Fs))):
            for _ in range(3):
                fn()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                    for _ in range(5):
                        fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s):
                g.replay()
                e0.record()
                for _ in range(iters):
                    g.replay()
                e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (iters * 5)
            out[name] = {"us": round(us, 1), "TFLOPs": round(flops / us * 1e-6, 1), "frac_mfma": round(flops / us * 1e-6 / 157.3, 3),
                         "GBps": round(nbytes / us * 1e-3, 1), "frac_hbm": round(nbytes / us * 1e-3 / HBM_PEAK_GBPS, 3)}
    return out


def spmm_hbm_record(iters=20):
    """The CSR SpMM on an HBM-RESIDENT working set: one rank's share of configs[4] (A_ui[U_r, :]: 250 000 user rows x
    1 000 000 item columns, 12.5 M edges, d = 128; the gathered item table is 512 MB = 2x the 256 MiB Infinity Cache) and
    its transpose (the backward's partial product: 512 MB written). HIP events on the launch stream. Fractions: of the 8 TB/s
    HBM peak by algorithmic bytes (gather-per-edge, no cache credit), and the kernel's fabric bytes (FETCH_SIZE x 2 +
    WRITE_SIZE of the committed PMC pass) / time against the 4.3-4.5 TB/s a purely random 512-byte-row gather reaches on
    this chip (profiles/r03_fetch_calibration.json)."""
    from mmssl_amd import ops, synth
    from mmssl_amd.graph import GraphPlan
    d, U_r, I = 128, 250_000, 1_000_000
    raw = synth.interaction_matrix(U_r, I, 12_500_000, seed=11)
    ui_r = synth.normalised_rows(raw)
    P = GraphPlan(ui_r)
    X = torch.randn(I, d, device="cuda")
    G = torch.randn(U_r, d, device="cuda")
    pmc, pmc_file, stale, verified = pmc_record(("r06_spmm_hbm_pmc.json", "r05_spmm_hbm_pmc.json", "r04_spmm_hbm_pmc.json"))
    if stale:
        pmc, pmc_file = None, pmc_file + " [STALE: csrc/graph.hip changed since the pass]"
    out = {"what": "configs[4] rank shape, d=128: A_ui[U_r,:] 250000 x 1000000, 12.5M edges; gathered table 512 MB (HBM resident)",
           "random_gather_ceiling_GBps": 4400.0,
           "traffic_source": ("profiles/%s (rocprofv3 --pmc pass of tools/spmm_hbm_pmc.py in its own run, tied to the kernel "
                              "sources by sha256)" % pmc_file) if pmc_file else None}
    rng = np.random.default_rng(0)
    with torch.no_grad():
        for name, tr, Xin, m in (("forward", False, X, ui_r), ("transpose", True, G, ui_r.T.tocsr())):
            for _ in range(3):
                Yd = ops.spmm(P, Xin, transpose=tr)
            torch.cuda.synchronize()
            # the launch that is timed below, checked: 2048 sampled output rows (+ the 8 longest) against a float64 CPU product
            rows = np.unique(np.concatenate([rng.choice(m.shape[0], 2048, replace=False), np.argsort(np.diff(m.indptr))[-8:]]))
            ref = np.asarray(m[rows].astype(np.float64) @ Xin.double().cpu().numpy())
            got = Yd[torch.from_numpy(rows).cuda()].double().cpu().numpy()
            rows_err = float(np.abs(got - ref).max() / np.abs(ref).max())
            del Yd
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.spmm(P, Xin, transpose=tr)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            by = synth.spmm_bytes(m, d)
            # (algorithmic bytes count a gathered row once per EDGE: with the table partly cached their rate can exceed the
            # HBM peak - `algorithmic_over_peak` is a ratio of the byte model, not a roofline fraction; the fabric figures are)
            rec = {"us": round(us, 1), "rows_vs_cpu": float("%.3g" % rows_err), "rows_checked": int(rows.shape[0]),
                   "algorithmic_bytes": int(by), "algorithmic_GBps": round(by / us * 1e-3, 1),
                   "algorithmic_over_peak": round(by / us * 1e-3 / HBM_PEAK_GBPS, 4),
                   "edge_layers_per_s": round(m.nnz / us * 1e6, 1)}
            if rows_err > 5e-6:
                rec["rows_check_failed"] = True
            t = (pmc or {}).get(name)
            if t and t.get("fabric_bytes"):
                rec["traffic"] = int(t["fabric_bytes"])
                rec["fabric_GBps"] = round(t["fabric_bytes"] / us * 1e-3, 1)
                rec["fabric_frac_of_8TBps"] = round(rec["fabric_GBps"] / HBM_PEAK_GBPS, 4)
                rec["fabric_over_random_gather_ceiling"] = round(rec["fabric_GBps"] / 4400.0, 3)
            out[name] = rec
    del P, X, G
    torch.cuda.empty_cache()
    return out


def _time_us(fn, iters, warm=3):
    """HIP events on the current stream around `iters` eager calls of fn() (after `warm` untimed ones)."""
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


def f_rows_record(a, step, raw, plans):
    """SURVEY 8f rows on the bench shape, each on its own (HIP events, eager launches, inputs resident):
      u_sim    Trainer.u_sim_calculation (main.py:283-298): [B, d] . [d, I] with the train-row mask and the row L2
               normalisation fused (ops.usim forward) - 5 calls per reference batch. HBM-bound on the [B, I] score write
               (+ its re-read by the scale pass): bytes = 2 * 4 B I + 4 (B + I) d.
      eval     one evaluation block of utility/batch_test.py:112-169 for 1024 users: scores with the training items at
               -inf + top-max(Ks)=50 per row + hit membership (ops.sim_rows / topk_rows / rows_membership).
      rebuild  the device-side modal-graph rebuild (main.py:378-405) from B * k (user, item) pairs: CSR + transpose + plans
               of both directions (graph.DeviceGraphPair.rebuild), k = int(I * m_topk_rate) as the reference computes it
               (1 for this shape), and the top-k selection that feeds it."""
    from mmssl_amd import ops
    from mmssl_amd.graph import DeviceGraphPair
    m = step.model
    U, I = raw.shape
    d = a.d
    B = a.batch
    dev = m.user_id_embedding.weight.device
    g = torch.Generator().manual_seed(11)
    ua = torch.randn(U, d, generator=g).to(dev)
    ia = torch.randn(I, d, generator=g).to(dev)
    users = torch.randperm(U, generator=g)[:B].to(dev)
    out = {}
    with torch.no_grad():
        us = _time_us(lambda: ops.usim(users, ua, ia, plans[0]), 200)
        ld = (I + 31) // 32 * 32
        by = 4.0 * B * ld + 4.0 * (B + I) * d          # ONE write of the pitched [B, ld] matrix + the two operand tables
        flop = 2.0 * B * I * d
        out["u_sim"] = {"what": "[%d, %d] masked, row-normalised scores (ops.usim forward): Gram-matrix row norms, then one "
                                "pass that writes the scaled matrix once (round 5: scores + partial sums, then a scale pass "
                                "over the matrix = 2 writes + 1 read, 107.6 us)" % (B, I), "us": round(us, 1),
                        "algorithmic_MB": round(by * 1e-6, 1), "GBps": round(by / us * 1e-3, 1),
                        "frac_hbm": round(by / us * 1e-3 / HBM_PEAK_GBPS, 3),
                        "frac_fp32_mfma": round(flop / us * 1e-6 / 157.3, 3)}
        # evaluation block: the train CSR as the mask (int32 rowptr / sorted cols on the device), positives = the train rows too
        rp = torch.from_numpy(raw.indptr.astype(np.int32)).to(dev)
        cols = torch.from_numpy(raw.indices.astype(np.int32)).to(dev)

        acc = torch.zeros((4, 8), dtype=torch.float64, device=dev)
        box = {"ws": None}

        def eval_block():
            rate, _ = ops.sim_rows(ua, ia, qidx=users, mask=(rp, cols), mask_value=float("-inf"))
            order = ops.topk_rows(rate, 50)
            box["ws"] = ops.eval_accumulate(rp, cols, users, order, [10, 20, 50], acc, box["ws"])
        us = _time_us(eval_block, 50)
        # end to end, as utility/batch_test.test_torch runs it: 16 blocks queued from the host loop, ONE read-back of the
        # metric sums at the end (wall clock, host overheads included)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(16):
            eval_block()
        acc.cpu()
        e2e = (time.perf_counter() - t0) / 16
        out["eval"] = {"what": "score + mask + top-50 + precision / recall / ndcg / hit @ 10, 20, 50 on the device for %d users "
                               "x %d items" % (B, I), "us": round(us, 1),
                       "ms_per_1k_users": round(us * 1e-3 * 1000.0 / B, 3), "users_per_s": round(B / us * 1e6, 1),
                       "end_to_end_ms_per_1k_users": round(e2e * 1e3 * 1000.0 / B, 3)}
        k = max(1, int(I * 0.0001))
        S = torch.randn(B, I, generator=g).to(dev)
        pair = DeviceGraphPair(U, I, DeviceGraphPair.MAX_PAIRS)

        def rebuild():
            ids = ops.topk_rows(S, k)
            pair.rebuild(users.repeat(k), ids.reshape(-1))
        us_all = _time_us(rebuild, 100)
        ids = ops.topk_rows(S, k)
        uu, ii = users.repeat(k), ids.reshape(-1)
        us_rb = _time_us(lambda: pair.rebuild(uu, ii), 100)
        out["modal_graph_rebuild"] = {"what": "top-%d of [%d, %d] scores, then CSR + transpose + plans of both directions from "
                                              "%d pairs, all on the device" % (k, B, I, B * k),
                                      "pairs": B * k, "us_topk_plus_rebuild": round(us_all, 1), "us_rebuild": round(us_rb, 1)}
        pair.destroy()
    return out


def first_step_loss(a, step, raw, batch):
    """HIP side of the loss check: the loss of ONE hot-path forward on the bench model's initial parameters with
    injected dropout masks. Returns (loss, snapshot of the inputs) — cpu_baseline() evaluates the CPU oracle on the
    snapshot (north_star: fp32 loss within 1e-4 relative)."""
    m = step.model
    I = raw.shape[1]
    gen = torch.Generator().manual_seed(123)
    km = [(torch.rand(I, a.d, generator=gen) >= 0.2) for _ in range(2)]
    users, pos, neg = (torch.from_numpy(x) for x in batch)
    step.keep_masks = tuple(k.to(torch.uint8).cuda() for k in km)
    step.set_batch(users.cuda(), pos.cuda(), neg.cuda())
    with torch.no_grad(), torch.cuda.stream(step.stream):
        total, _ = step.losses()
    torch.cuda.synchronize()
    hip = float(total)
    step.keep_masks = None
    P = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()
         if not k.startswith(("image_embedding", "text_embedding", "batch_norm", "encoder.", "align."))}
    return hip, {"P": P, "img": m.image_feats.cpu(), "txt": m.text_feats.cpu(), "keep": [k.float() for k in km],
                 "batch": (users, pos, neg)}


TRAFFIC_FILE = [None]
SPMM_SOURCES = ("mmssl_amd/csrc/graph.hip", "mmssl_amd/csrc/graph_internal.hpp", "mmssl_amd/csrc/common.hpp")


def kernel_source_sha(files=SPMM_SOURCES):
    """sha256 over the sources of a kernel: tools/pmc_summary.py records it next to the counters of a PMC pass, and a
    `traffic` figure is only printed while the kernel it was measured on is still the kernel in the tree."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def pmc_record(names, key=None):
    """(record, file name, stale?) of the newest committed PMC summary among `names`. A summary that carries a
    `source_sha256` other than the current sources' is STALE (its numbers describe another kernel); summaries from before
    round 6 carry none and count as unverified-but-usable only while no verified one exists."""
    cur = kernel_source_sha()
    for name in names:
        p = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(p):
            continue
        try:
            rec = json.load(open(p))
        except Exception:
            continue
        sha = rec.get("source_sha256")
        return rec, name, (sha is not None and sha != cur), sha is not None
    return None, None, False, False


def load_traffic():
    """Fabric bytes per SpMM launch from the committed rocprofv3 PMC pass (profiles/*_spmm_pmc.json); None when the pass
    was taken on other kernel sources than the tree's (re-run `bash tools/evidence.sh rNN pmc`)."""
    rec, name, stale, verified = pmc_record(("r06_spmm_pmc.json", "r05_spmm_pmc.json", "r04_spmm_pmc.json"))
    if rec is None:
        return None
    TRAFFIC_FILE[0] = name + ("" if verified else " [no source hash recorded]")
    if stale:
        TRAFFIC_FILE[0] = name + " [STALE: csrc/graph.hip changed since the pass - figure withheld]"
        return None
    return rec.get("hbm_bytes_per_launch")


def cpu_baseline(a, raw, mats, budget_s=20.0, first=None):
    """The CPU oracle (torch-CPU restatement of the reference path, 'port') on this box's host cores:
    the same step (forward + losses + backward; torch COO sparse.mm like the reference), a bounded
    number of iterations. With `first` = first_step_loss()'s result the oracle also evaluates that very step
    (same parameters, masks, batch) and the two losses are reported side by side (`loss_check`)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mmssl_oracle as O
    from mmssl_amd import synth
    import scipy.sparse as sp
    U, I, E, dv, dt = synth.SHAPES[a.workload]
    # torch-CPU sparse/elementwise kernels stop scaling (and collapse) far below the 256 hardware
    # threads of the GPU box: 32 threads is the measured sweet spot class for this path
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ui, iu = mats
    A_ui, A_iu = O.to_torch_sparse(ui).coalesce(), O.to_torch_sparse(iu).coalesce()
    e_ui = O.to_torch_sparse(sp.csr_matrix((U, I), dtype=np.float32))
    e_iu = O.to_torch_sparse(sp.csr_matrix((I, U), dtype=np.float32))
    graphs = (A_ui, A_iu, e_ui, e_iu, e_ui, e_iu)
    g = torch.Generator().manual_seed(7)
    d = a.d
    img, txt = torch.randn(I, dv, generator=g), torch.randn(I, dt, generator=g)
    P = {"image_trans.weight": torch.randn(d, dv) * 0.02, "image_trans.bias": torch.zeros(d),
         "text_trans.weight": torch.randn(d, dt) * 0.02, "text_trans.bias": torch.zeros(d),
         "user_id_embedding.weight": torch.randn(U, d) * 0.01, "item_id_embedding.weight": torch.randn(I, d) * 0.01,
         "weight_dict.w_q": torch.randn(d, d) * 0.1, "weight_dict.w_self_attention_cat": torch.randn(4 * d, d) * 0.1}
    for v in P.values():
        v.requires_grad_(True)
    cfg = O.Cfg(embed_size=d, n_ui_layers=a.gcn_layers, layers=1, drop_rate=0.2, batch_size=a.batch)
    check = None
    if first is not None:
        hip, snap = first
        u_, p_, n_ = snap["batch"]
        with torch.no_grad():
            o = O.forward(snap["P"], snap["img"], snap["txt"], graphs, cfg, training=True, keep_masks=snap["keep"])
            mf, emb, _ = O.bpr(o[0][u_], o[1][p_], o[1][n_], 1e-5, a.batch)
            ref = float(mf + emb + O.feat_reg(o[2], o[3], o[4], o[5], I, 1e-5) + 0.03 * (
                O.infonce(o[8][u_], o[6][u_], 0.5) + O.infonce(o[9][u_], o[6][u_], 0.5)))
        check = {"hip": round(hip, 7), "oracle": round(ref, 7), "rel_err": float("%.3g" % (abs(hip - ref) / abs(ref))),
                 "tolerance": 1e-4, "what": "loss of the first step (initial parameters, injected dropout masks)"}
        del o, snap
    users, pos, neg = (torch.from_numpy(x) for x in make_batches(raw, 1, a.batch, 3)[0])
    keep = [(torch.rand(I, d) >= 0.2).float() for _ in range(2)]
    n_spmm = 2 * (4 + 2 * a.gcn_layers)     # nonzero-graph launches, forward + backward
    opt = torch.optim.AdamW(list(P.values()), lr=5.5e-4)       # the generator's optimiser (main.py:76-80): a full step

    def step():
        for v in P.values():
            v.grad = None
        o = O.forward(P, img, txt, graphs, cfg, training=True, keep_masks=keep)
        mf, emb, _ = O.bpr(o[0][users], o[1][pos], o[1][neg], 1e-5, a.batch)
        loss = mf + emb + O.feat_reg(o[2], o[3], o[4], o[5], I, 1e-5) + 0.03 * (
            O.infonce(o[8][users], o[6][users], 0.5) + O.infonce(o[9][users], o[6][users], 0.5))
        loss.backward()
        opt.step()
    t0 = time.time()
    step()                                   # warm-up (also bounds the sample if the host is slow)
    first = time.time() - t0
    t0 = time.time()
    n = 0
    while n == 0 or (time.time() - t0 < budget_s - first and n < 20):
        step()
        n += 1
    dt_s = (time.time() - t0) / n
    # the propagation alone, on torch's CSR kernels (the CPU's best case; the reference itself uses COO)
    gcn = {}
    with torch.no_grad():
        Eu, Ei = torch.randn(U, d), torch.randn(I, d)
        for fmt, (Au, Ai) in (("coo", (A_ui, A_iu)), ("csr", (A_ui.to_sparse_csr(), A_iu.to_sparse_csr()))):
            for threads in sorted({min(32, os.cpu_count() or 1), os.cpu_count() or 1}):
                torch.set_num_threads(threads)
                t0, k = time.time(), 0
                while k < 2 or (time.time() - t0 < 1.5 and k < 50):
                    u, i = Eu, Ei
                    for _ in range(a.gcn_layers):
                        u = torch.sparse.mm(Au, i)
                        i = torch.sparse.mm(Ai, u)
                    k += 1
                gcn["%s_%dthr" % (fmt, threads)] = round(2 * a.gcn_layers * raw.nnz / ((time.time() - t0) / k), 1)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    # the evaluation's CPU form (batch_test.py:21-36, 112-169: score GEMM, then per user a heap selection over the
    # non-training items + the metric formulas; the reference forks a Pool over cores // 5 workers): 128 users, one process
    eval_cpu = None
    try:
        gq = torch.Generator().manual_seed(11)
        ua_c, ia_c = torch.randn(U, d, generator=gq), torch.randn(I, d, generator=gq)
        us_c = torch.randperm(U, generator=gq)[:128].tolist()
        tr_items = {u: raw.indices[raw.indptr[u]:raw.indptr[u + 1]].tolist() for u in us_c}
        t0 = time.time()
        O.evaluate(ua_c, ia_c, us_c, tr_items, {u: set(tr_items[u][:1]) for u in us_c}, [10, 20, 50])
        el = time.time() - t0
        eval_cpu = {"ms_per_1k_users": round(el * 1e3 * 1000.0 / len(us_c), 1), "users": len(us_c), "processes": 1,
                    "kind": "port (oracle.evaluate: torch-CPU scores + per-user stable ranking)"}
    except Exception as e:       # never lose the line over the side record
        eval_cpu = {"error": repr(e)[:200]}
    return {"value": round(n_spmm * raw.nnz / dt_s, 1), "unit": "edge.layers/s", "cores": torch.get_num_threads(),
            "eval_cpu": eval_cpu,
            "host_cores": os.cpu_count(), "gcn_forward_edge_layers_per_s": gcn, "loss_check": check,
            "kind": "port", "ms_per_step": round(dt_s * 1e3, 1),
            "sample": "%d steps of the same %s-shape step (fwd+losses+bwd+AdamW) by oracle/mmssl_oracle.py "
                      "on torch-CPU COO sparse.mm" % (n, a.workload)}


def graph_capture_works(a):
    """A failed hipGraph capture can take the process down (runtime abort), so it is tried in a
    child process on a small shape first; on failure the bench falls back to eager launches."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--graph-probe", "--workload", "tiktok", "--d", str(a.d),
           "--gcn-layers", str(a.gcn_layers), "--batch", str(a.batch)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        return r.returncode == 0
    except Exception:
        return False


def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` WITHOUT a launcher: start the N ranks ourselves - the same command line under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1, a free port) - and pass its exit code on.
    Rank 0's JSON line reaches this process's stdout through the launcher."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(n, 1))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def launch_check(a, rank, world, local_rank):
    """Dry run of the launch path: every rank joins the process group (gloo works without a GPU), one all-reduce and the
    barrier / max-over-ranks timing skeleton of the real run; rank 0 prints ONE JSON line."""
    import torch.distributed as dist
    use_cuda = a.backend == "nccl"
    if use_cuda:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank) if use_cuda else torch.device("cpu")
    dist.init_process_group(a.backend, **({"device_id": dev} if use_cuda else {}))
    dist.barrier()
    t0 = time.perf_counter()
    t = torch.tensor([float(rank + 1)], device=dev, dtype=torch.float64)
    dist.all_reduce(t)
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    ok = float(t.item()) == world * (world + 1) / 2
    if rank == 0:
        print(json.dumps({"launch_check": bool(ok), "n_gpus": world, "backend": a.backend,
                          "rank_sum": float(t.item()), "max_elapsed_s": round(float(el.item()), 4)}))
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="baby")
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--gcn-layers", type=int, default=3, dest="gcn_layers")
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--stress-timeout", type=int, default=300,
                    help="N > 1: seconds the scaling_stress run may take before the line is printed without it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N>1: weak = shape x N (per-rank work fixed); strong = the shape itself cut N ways")
    ap.add_argument("--only", choices=["all", "steps", "roofline"], default="all",
                    help="profiling aid: run only the timed steps, or only the isolated SpMM roofline loop")
    ap.add_argument("--dense-fuse", action="store_true", dest="dense_fuse",
                    help="A/B: the dense two-sided fuse launch in front of the loss chain (default: batch rows only, the "
                         "regulariser sums on the side stream)")
    ap.add_argument("--no-fuse-adam", action="store_true", dest="no_fuse_adam",
                    help="A/B aid: one AdamW launch after the backward instead of the fused / side-stream updates")
    ap.add_argument("--graph-probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dist-graph", choices=["auto", "on", "off"], default="auto", nargs="?", const="on",
                    help="sharded path: capture the step (RCCL collectives included) into a hipGraph. auto = "
                         "only if a small-shape capture+replay succeeds on every rank in child processes")
    ap.add_argument("--dist-graph-probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--force-dist", action="store_true",
                    help="use the row-sharded code path even with one rank (exercises RCCL + dist.py on 1 GPU)")
    ap.add_argument("--launch-check", action="store_true", dest="launch_check",
                    help="launcher dry run (no GPU needed): rendezvous of --gpus ranks on --backend, one all-reduce, one "
                         "JSON line from rank 0")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="process-group backend (gloo: --launch-check only)")
    ap.add_argument("--graph", choices=["uniform", "communities"], default="uniform",
                    help="N=1: uniform = the SURVEY 8d generator (no column locality; the metric's workload); communities = "
                         "the same shape as 8 user / item communities with 10 %% global edges (what the XCD-banded SpMM work "
                         "list is for)")
    ap.add_argument("--xcd-bands", type=int, default=0, dest="xcd_bands", choices=[-1, 0, 1],
                    help="GraphPlan XCD banding: 0 = automatic (on when the graph has column locality), 1 = always, -1 = never")
    ap.add_argument("--proj", choices=["split", "f32"], default="split",
                    help="grouped projection kernels: split = exact 3-way bf16 cut, six partial products on the bf16 matrix pipe "
                         "(fp32-accurate, default); f32 = the fp32-MFMA kernels (A/B)")
    ap.add_argument("--no-frows", action="store_true", dest="no_frows", help="skip the `f_rows` record (u_sim / evaluation / "
                    "modal-graph rebuild timings)")
    ap.add_argument("--no-hbm", action="store_true", dest="no_hbm", help="skip the HBM-resident SpMM record (`spmm_hbm`)")
    ap.add_argument("--share-gpu", action="store_true", dest="share_gpu",
                    help="test aid: every rank on GPU 0, process group on gloo moving device tensors (RCCL refuses two ranks on "
                         "one device) - runs the whole N > 1 flow on a one-GPU box; eager steps, timings mean nothing")
    ap.add_argument("--no-stress", action="store_true", dest="no_stress",
                    help="N>1: skip the `scaling_stress` record (configs[4]'s per-rank share x N after the timed region)")
    ap.add_argument("--transport", choices=["auto", "peer", "collective"], default="auto",
                    help="N > 1: how table rows move between the ranks. peer = kernels writing / reading IPC-mapped peer "
                         "windows (csrc/peer.hip, no collective library in the data path); collective = RCCL (gloo with "
                         "--share-gpu); auto = peer if its start-up self-test passes on every rank, else collective")
    ap.add_argument("--peer-sabotage", action="store_true", dest="peer_sabotage", help=argparse.SUPPRESS)   # tests: see choose_transport
    ap.add_argument("--no-loss-check", action="store_true", dest="no_loss_check",
                    help="N > 1: skip `loss_vs_n1` (the sharded job's first-step loss against the same job whole on rank 0)")
    ap.add_argument("--scheme", choices=["item-side", "gather-both", "halo"], default="item-side",
                    help="sharded step: item-side = user-row blocks only, every collective of item-table size; "
                         "gather-both = user AND item row blocks, all-gather of both tables (round-3 scheme); halo = item-side "
                         "exchanging only the item rows a rank's edges reference (all-to-all of row lists)")
    ap.add_argument("--replicate-feats", choices=["auto", "on", "off"], default="auto", dest="replicate_feats",
                    help="item-side scheme: keep the whole constant feature matrices on every rank and project all items "
                         "locally, so the projected features never travel (auto: by dist.choose_replicate_feats - on for "
                         "configs[4]'s narrow features, off for the Baby shape)")
    ap.add_argument("--chunks", type=int, default=0,
                    help="sharded step: column chunks per collective (chunk c's SpMM runs under chunk c+1's collective); "
                         "0 = by size (1 below 64 MB per collective, else 2-4)")
    a = ap.parse_args()
    if a.proj == "f32":
        from mmssl_amd import ops as _ops
        _ops.PROJ_SPLIT = False
    if a.workload in ("synth", "synth-full") and a.d == 64:
        a.d = 128                      # configs[4] is defined at d=128
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but the launcher started %d rank(s)" % (a.gpus, world))
    import torch.distributed as dist
    if a.launch_check:
        launch_check(a, rank, world, local_rank)
        return
    if a.share_gpu:
        local_rank = 0
        a.backend = "gloo"          # (the gloo stand-in cannot be captured: run_sharded_main turns the graph off for it)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or a.force_dist or a.workload in ("synth", "synth-full") or a.dist_graph_probe
    if sharded:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:          # a free port: back-to-back runs must not collide
                os.environ["MASTER_PORT"] = str(free_port())
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        elif a.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if sharded:
        run_sharded_main(a, rank, world, dev)
        return
    step, raw, mats, plans = build_single_gpu(a, dev)
    stats = count_edge_layers(step, a.d)
    if a.graph_probe:            # child process: does whole-step hipGraph capture + replay work here?
        ok = step.capture()
        if ok:
            step.run()
            torch.cuda.synchronize()
        sys.exit(0 if ok else 3)
    first = None
    if a.only == "all" and not a.no_cpu_baseline:
        first = first_step_loss(a, step, raw, make_batches(raw, 1, a.batch, seed=5)[0])
    batches = [torch.stack([torch.from_numpy(x).to(dev) for x in b])      # packed [3, B]
               for b in make_batches(raw, 8, a.batch, seed=2022)]
    # the eight batches stay resident as a ring: every step picks slot (completed steps mod 8) by a launch of its own
    # (HotPathStep.set_batch_ring; BEFORE the capture: that launch is part of the step), so two replays are not
    # separated by a host-issued copy
    step.set_batch_ring(torch.stack(batches))
    captured = (not a.no_graph) and graph_capture_works(a) and step.capture()
    edge_layers_total = stats["edge_layers"]

    def run_steps(n):
        for _ in range(n):
            step.run()

    if a.only == "roofline":
        a.warmup, a.steps = 1, 1
    # the isolated-kernel records first (they also bring the clocks up before the short timed region)
    extra = {}
    if a.only != "steps":
        extra["roofline"] = spmm_roofline(plans, mats, a.d)
        extra["gcn_forward"] = gcn_forward_record(plans, mats, a.d, a.gcn_layers)
        extra["projection"] = projection_record(step)
    run_steps(a.warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(a.steps)
    t_enq = time.perf_counter() - t0            # the host's share: all K step.run() calls have returned
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = elapsed * 1e3 / a.steps
    out = result_line(a, 1, "weak", ms, edge_layers_total, stats, captured, float(step.loss), "single",
                      int(raw.shape[0]), int(raw.shape[1]), int(raw.nnz))
    # host time inside step.run() per step (one hipGraphLaunch when captured). Far below ms_per_step = the GPU is the
    # bound and the host runs ahead; equal to it = the host (or a full queue) paces the step.
    out["host_enqueue_us"] = round(t_enq * 1e6 / a.steps, 1)
    info = plans[0].info()
    out["config"]["graph"] = "%s; XCD-banded work list: %s (locality score %.2f / %.2f)" % (
        a.graph, "on" if info["banded"] else "off", info["band_score"], plans[1].info()["band_score"])
    if a.only != "steps":
        rf = extra["roofline"]
        # the same byte model over the WHOLE step: every SpMM launch's algorithmic bytes / step time. The
        # step is not SpMM-bound (projection GEMMs and the loss section share it), so this is far lower.
        rf["step_spmm_GBps"] = round(stats["spmm_bytes"] / (ms * 1e-3) * 1e-9, 1)
        rf["step_frac"] = round(rf["step_spmm_GBps"] / HBM_PEAK_GBPS, 4)
        out["roofline"] = rf
        out["gcn_forward"] = extra["gcn_forward"]
        out["projection"] = extra["projection"]
        if a.only == "all" and not a.no_hbm:
            # AFTER the timed region: building the 12.5 M-edge graph takes seconds of host time, during which the GPU
            # clocks fall back - in front of the short timed region that cost the driver's 20-step command 5-10 %
            out["spmm_hbm"] = spmm_hbm_record()
    if a.only == "all" and not a.no_frows:
        out["f_rows"] = f_rows_record(a, step, raw, plans)
    if not a.no_cpu_baseline and a.only == "all":
        out["cpu_baseline"] = cpu_baseline(a, raw, mats, first=first)
        out["loss_check"] = out["cpu_baseline"].pop("loss_check")
        ev = out["cpu_baseline"].pop("eval_cpu", None)
        if ev and "f_rows" in out:
            out["f_rows"]["eval"]["cpu"] = ev
    print(json.dumps(out))


def result_line(a, world, scaling, ms, edge_layers_total, stats, captured, loss, parallelism, n_users, n_items, n_edges):
    shape_note = "" if world == 1 else (" x%d (weak: per-rank share fixed)" % world if scaling == "weak"
                                        else " cut %d ways (strong)" % world)
    return {
        "metric": "edge.layers/s (d-wide multiply-adds per nonzero over every SpMM launch of a hot-path step / step time)",
        "value": round(edge_layers_total / (ms * 1e-3), 1), "unit": "edge.layers/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s-shaped graph%s, d=%d, %d-layer GCN + V/T projection + InfoNCE x2 + BPR + "
                               "feat-reg, fwd+bwd+AdamW, B=%d" % (a.workload, shape_note, a.d, a.gcn_layers, a.batch),
                   "n_users": n_users, "n_items": n_items, "n_edges": n_edges,
                   "edge_layers_per_step": int(edge_layers_total), "spmm_launches_per_step": int(stats["spmm_launches"]),
                   "launch": "hipGraph replay" if captured else "eager", "parallelism": parallelism,
                   "final_loss": round(loss, 6)},
    }


def timed_sharded(a, rank, world, dev, scaling, want_graph, pre_step=None):
    """Build the row-sharded step for workload `a`, W warm-up steps, then EXACTLY K timed steps between barrier +
    synchronize on both sides, max over ranks. Returns a dict (ms, stats, comm, ...)."""
    import torch.distributed as dist
    from mmssl_amd import dist as mdist
    step, mats, plans, stats = mdist.build_bench_step(a, rank, world, dev, scaling, pre_step=pre_step)
    captured = bool(want_graph and step.capture())
    n_users, n_items = stats["n_users"], stats["n_items"]
    # communication of one step: what was issued, and how long those collectives take on their own
    log = stats["comm_log"]
    comm = {"scheme": stats["scheme"], "replicate_feats": stats.get("replicate_feats", False), "column_chunks": stats["chunks"],
            "collectives_per_step": len(log),
            "bytes_per_step": int(sum(b for _, _, b in log)),
            "by_kind": {k: [sum(1 for x in log if x[0] == k), int(sum(x[2] for x in log if x[0] == k))]
                        for k in ("all_gather", "reduce_scatter", "all_reduce", "halo_gather", "halo_reduce", "peer_gather",
                                  "peer_reduce")
                        if k in ("all_gather", "reduce_scatter", "all_reduce") or any(x[0] == k for x in log)},
            "note": "bytes = size of the full (gathered / to-be-scattered / reduced) fp32 buffer of every collective "
                    "of one step on one rank; comm_only_ms = the same collectives replayed alone, back to back"}
    pc = mdist._peer(None)
    if pc is None:
        with torch.cuda.stream(step.stream):
            comm["comm_only_ms"] = round(mdist.comm_replay_ms(log, None, dev, halo=stats.get("halo")), 4)
    else:
        comm["comm_only_ms"] = None
    # what one step hands to torch.distributed in its data path (an eager step; the count is 0 over the peer exchange)
    before = mdist.COMM["dist_calls"]
    step.step()
    torch.cuda.synchronize()
    comm["torch_distributed_collectives_per_step"] = mdist.COMM["dist_calls"] - before
    if pc is not None:
        l0 = pc.t.launches
        step.step()
        peer_agree(pc, dev, "first eager steps")
        st = pc.stats()
        comm["peer"] = {"exchange_call_sites_per_step": st["call_sites"], "exchange_launches_per_step": pc.t.launches - l0,
                        "windows": st["windows"], "window_MB": round(st["window_bytes"] / 2 ** 20, 1),
                        "flags_finegrained": st["flags_finegrained"]}
    if stats.get("halo_rows_fraction") is not None:
        comm["halo_rows_fraction"] = stats["halo_rows_fraction"]      # referenced item rows / item table (this rank)
    rngb = np.random.default_rng(2022)          # identical batches on every rank (global ids)
    batches = [torch.stack([torch.from_numpy(x).to(dev) for x in (                   # packed [3, B]: one copy per step
        rngb.choice(n_users, a.batch, replace=a.batch > n_users).astype(np.int64),
        rngb.integers(0, n_items, a.batch).astype(np.int64), rngb.integers(0, n_items, a.batch).astype(np.int64))])
        for _ in range(8)]

    def run_steps(n):
        for i in range(n):
            step.set_batch(batches[i % len(batches)])
            step.run()
    run_steps(a.warmup)
    if pc is not None:
        peer_agree(pc, dev, "warm-up steps")
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(a.steps)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if pc is not None:
        peer_agree(pc, dev, "timed steps")          # a wait that gave up inside the timed region voids the figure
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = float(t.item()) * 1e3 / a.steps
    return {"ms": ms, "stats": stats, "comm": comm, "captured": captured, "loss": float(step.loss), "plans": plans,
            "mats": mats, "step": step}


def committed_rank_figure():
    """ms/step and edge.layers/s of configs[4]'s per-rank share on ONE rank (no link crossed), from the committed run."""
    for name in ("r06_bench_synth_w1.json", "r05_bench_synth_w1.json", "r04_bench_synth_w1.json", "r03_bench_synth_w1.json"):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            try:
                with open(p) as f:
                    d = json.loads([ln for ln in f.read().splitlines() if ln.startswith("{")][0])
                return {"file": "profiles/" + name, "ms_per_step": d["ms_per_step"], "edge_layers_per_s": d["value"]}
            except Exception:
                pass
    return None


def committed_full_n1_figure():
    """ms/step of configs[4] WHOLE (2M x 1M x 100M edges, d = 128) on ONE GPU (`--workload synth-full`), from the committed
    run: the denominator of north_star's ">= 6x 1 -> 8" on the shape where that is arithmetically possible."""
    for name in ("r06_bench_synth_full_n1.json", "r05_bench_synth_full_n1.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.loads([ln for ln in f.read().splitlines() if ln.startswith("{")][0])
            return {"file": "profiles/" + name, "ms_per_step": d["ms_per_step"], "edge_layers_per_s": d["value"]}
        except Exception:
            pass
    return None


def stress_watchdog(limit_s, out, rank):
    """The N > 1 line must survive a scaling_stress run that HANGS (a collective some rank never joins cannot be caught
    as an exception): after `limit_s` seconds rank 0 prints the headline line with the time-out noted and every rank
    leaves with rc 0. Returns the started timer; cancel() it when the stress run is over."""
    import threading

    def fire():
        if rank == 0:
            out["scaling_stress"] = {"error": "timed out after %d s (run abandoned; the headline figures above were "
                                              "complete before it started)" % limit_s}
            print(json.dumps(out), flush=True)
        os._exit(0)
    t = threading.Timer(limit_s, fire)
    t.daemon = True
    t.start()
    return t


def peer_selftest(pc, dev, rounds=4):
    """All-gather / reduce-scatter / all-reduce round trips through the peer windows on values every rank knows - `rounds`
    times THROUGH THE SAME WINDOWS with different values each time, every result read by ordinary (cached) kernels: besides
    the plumbing this checks, on the devices the job really runs on, what the exchange's visibility protocol assumes -
    that a kernel launched behind the wait sees the rows the peers have just written into a window it read (and cached) one
    round earlier. A stale read shows up here as a mismatch in round >= 2 and sends the job to the collective transport."""
    w, r, per, width = pc.world, pc.rank, 512, 64
    ramp = torch.arange(width, device=dev, dtype=torch.float32)
    ok = True
    for k in range(rounds):
        pc.begin_step()                                   # (call order restarts: the same windows as the round before)
        base = float(1 + 3 * k)
        want = torch.cat([torch.full((per, width), base * (q + 1), device=dev) + ramp for q in range(w)])
        full = pc.gather(want[r * per:(r + 1) * per].contiguous())
        ok = ok and bool(torch.equal(full, want))
        P = pc.partial(w * per, width)
        P.copy_(want * float(r + 1))
        tot = float(sum(q + 1 for q in range(w)))
        ok = ok and bool(torch.allclose(pc.reduce(P, per), want[r * per:(r + 1) * per] * tot, rtol=1e-6))
        t = torch.arange(1001, device=dev, dtype=torch.float32) * (base * (r + 1))
        pc.all_reduce_(t)
        ok = ok and bool(torch.allclose(t, torch.arange(1001, device=dev, dtype=torch.float32) * (base * tot), rtol=1e-6))
    torch.cuda.synchronize()
    pc.check()
    return ok


class PeerFailed(RuntimeError):
    """Raised by EVERY rank at the same point (peer_agree) when the peer exchange failed on any of them."""


def peer_agree(pc, dev, where):
    """An agreed health check of the peer exchange: every rank reads its context's error word, the ranks all-reduce the
    outcome over torch.distributed, and either all go on or all raise PeerFailed - a rank whose waits timed out must not
    leave the others alone in the next collective."""
    import torch.distributed as dist
    bad, msg = 0, ""
    try:
        torch.cuda.synchronize()
        pc.check()
    except Exception as e:          # noqa: BLE001
        bad, msg = 1, repr(e)[:200]
    t = torch.tensor([bad], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if int(t.item()):
        raise PeerFailed("%s: %s" % (where, msg or "a wait timed out on another rank"))


def choose_transport(a, rank, world, dev):
    """(name, record). `auto`: the peer exchange if it can be set up and its self-test passes on EVERY rank."""
    import torch.distributed as dist
    from mmssl_amd import dist as mdist
    if world == 1 and not a.force_dist:
        return "none", {"transport": "none (one rank: every exchange is the identity)"}
    if a.transport == "collective" or a.scheme == "halo" or (world == 1 and a.transport == "auto"):
        return "collective", {"transport": "collective (%s)" % dist.get_backend()}
    ok, err = 1, None
    try:
        pc = mdist.enable_peer_exchange(None, dev, timeout_ms=30000)
        ok = 1 if peer_selftest(pc, dev) else 0
        if not ok:
            err = "self-test values differ"
    except Exception as e:          # IPC export / open refused, a wait timed out, ...
        ok, err = 0, repr(e)[:300]
    flag = torch.tensor([ok], dtype=torch.int32, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        pc = mdist._peer(None)
        if a.peer_sabotage:
            # test hook (tests/test_bench_contract_gpu.py): after a PASSED self-test the last rank stops signalling its
            # reduce-scatter partials - the other ranks' waits give up after 0.5 s, the agreed check raises PeerFailed on
            # every rank and the job falls back to collectives
            pc.t._L.mmssl_peer_set_timeout_ms(pc.t._ctx, 500)
            if rank == world - 1:
                real, only_wait = pc.t.signal_wait, pc.t.wait
                pc.t.signal_wait = lambda ch: real(ch) if ch == 0 else only_wait(ch)
        info = pc.stats()
        return "peer", {"transport": "peer (IPC-mapped windows + epoch flags, csrc/peer.hip)",
                        "flags_finegrained": info["flags_finegrained"],
                        "selftest": "passed on all %d ranks (4 rounds through the same windows, cached readers)" % world}
    if a.transport == "peer":
        raise SystemExit("--transport peer: the peer exchange is not usable here (rank %d: %s)" % (rank, err))
    try:
        mdist.disable_peer_exchange(None)
    except Exception:
        pass
    return "collective", {"transport": "collective (%s)" % dist.get_backend(),
                          "peer_rejected": err or "another rank's self-test failed"}


def rank_devices(world, dev):
    """Which physical devices the job's ranks sit on: UUIDs gathered from every rank (a job whose ranks share a device, or
    a launcher that started fewer processes than --gpus says, shows here)."""
    import torch.distributed as dist
    try:
        uid = str(torch.cuda.get_device_properties(dev).uuid)
    except Exception:
        uid = "%s:%d" % (torch.cuda.get_device_name(dev), dev.index or 0)
    got = [None] * world
    dist.all_gather_object(got, uid)
    return {"world_size": world, "backend": dist.get_backend(), "device_uuids": got, "distinct_devices": len(set(got))}


def loss_vs_n1(a, step, rank, world, dev, scaling, group1):
    """First-step loss of the sharded job (its own step object, injected dropout masks, batch 0) against the SAME job whole
    on rank 0's GPU (dist.build_whole_bench_step: one rank, no exchange): north_star's 1e-4."""
    import torch.distributed as dist
    from mmssl_amd import dist as mdist
    m = step.model
    rngb = np.random.default_rng(4242)
    n_users, n_items = m.ush.n, m.ish.n
    batch = torch.stack([torch.from_numpy(x) for x in (
        rngb.choice(n_users, a.batch, replace=a.batch > n_users).astype(np.int64),
        rngb.integers(0, n_items, a.batch).astype(np.int64), rngb.integers(0, n_items, a.batch).astype(np.int64))])
    if getattr(m, "replicate_feats", False):
        km = [mdist.bench_check_masks(q, m.ish.per, a.d) for q in range(world)]
        keep = tuple(torch.cat([k[i] for k in km]).to(dev) for i in range(2))
    else:
        keep = tuple(k.to(dev) for k in mdist.bench_check_masks(rank, m.ish.per, a.d))
    old = step.keep_masks
    step.keep_masks = keep
    with torch.cuda.stream(step.stream):
        step.set_batch(batch.to(dev))
        ln = float(step.backward())
    torch.cuda.synchronize()
    step.keep_masks = old
    rec = {"sharded": ln, "tolerance": 1e-4}
    if rank == 0:
        try:
            whole = mdist.build_whole_bench_step(a, world, dev, scaling, group1)
            with torch.cuda.stream(whole.stream):
                whole.set_batch(batch.to(dev))
                l1 = float(whole.backward())
            torch.cuda.synchronize()
            rec.update(n1=l1, rel_err=abs(ln - l1) / abs(l1), ok=bool(abs(ln - l1) <= 1e-4 * abs(l1)),
                       what="loss of one step from the initial parameters, injected dropout masks: the job on %d rank(s) vs "
                            "the same graph / tables / features whole on rank 0's GPU" % world)
            del whole
        except Exception as e:
            rec["error"] = repr(e)[:300]
    dist.barrier()
    return rec


def one_rank_ms(a, rank, dev, group1):
    """ms/step of workload `a`'s per-rank share ALONE on rank 0 (a one-rank group: every exchange is the identity), timed in
    this invocation like the job itself; the other ranks wait. Returned on every rank (broadcast as a python object)."""
    import gc
    import torch.distributed as dist
    from mmssl_amd import dist as mdist
    ms = None
    if rank == 0:
        saved = mdist._PEER.copy()
        mdist._PEER.clear()                  # (keyed by group, but build_sharded_graph's helpers use the default group)
        try:
            step, _, _, _ = mdist.build_bench_step(a, 0, 1, dev, "weak", group=group1)
            step.capture()
            for _ in range(max(a.warmup, 2)):
                step.run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step.run()
            torch.cuda.synchronize()
            ms = round((time.perf_counter() - t0) * 1e3 / a.steps, 4)
            del step
            gc.collect()
        finally:
            mdist._PEER.update(saved)
    box = [ms]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def run_sharded_main(a, rank, world, dev):
    import copy
    import gc
    import torch.distributed as dist
    from mmssl_amd import dist as mdist
    scaling = "weak" if a.workload in ("synth", "synth-full") else a.scaling
    group1 = dist.new_group([0]) if world > 1 else None          # (collective call: rank 0's one-rank group for loss_vs_n1)
    tname, trec = choose_transport(a, rank, world, dev) if not a.dist_graph_probe else ("collective", {})
    if tname == "peer":
        a.dist_graph = "on" if a.dist_graph == "auto" else a.dist_graph     # plain kernels: nothing a capture could choke on
    elif a.share_gpu:
        a.dist_graph = "off"
    if a.dist_graph_probe:       # child of one rank: sharded capture + replay on a small shape
        step, _, _, _ = mdist.build_bench_step(a, rank, world, dev, scaling)
        ok = step.capture()
        if ok:
            for _ in range(3):
                step.run()
            torch.cuda.synchronize()
            ok = bool(torch.isfinite(step.loss).item())
        dist.barrier()
        # the captured graph goes first: destroying the process group while a live hipGraph still holds its all-to-all
        # (send / recv) nodes hangs inside RCCL (seen with the halo scheme)
        del step
        gc.collect()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        sys.exit(0 if ok else 3)
    want = a.dist_graph != "off" and not a.no_graph
    if want and a.dist_graph == "auto":
        # A failed capture with RCCL inside can abort or hang the process: try it first in one child
        # per rank (own rendezvous on MASTER_PORT+1), then agree on the outcome across ranks.
        cmd = [sys.executable, os.path.abspath(__file__), "--dist-graph-probe", "--gpus", str(a.gpus),
               "--workload", "tiktok", "--d", str(a.d), "--gcn-layers", str(a.gcn_layers), "--batch", str(a.batch),
               "--scaling", "strong", "--scheme", a.scheme, "--chunks", str(max(a.chunks, 2) if world > 1 else a.chunks)]
        if a.force_dist:
            cmd.append("--force-dist")
        flag = torch.tensor([1 if mdist.spawn_rank_probe(cmd) else 0], device=dev, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        want = bool(flag.item())
    lbox = {}
    check = None
    if world > 1 and not a.no_loss_check and a.workload != "synth":
        check = lambda st: lbox.update(rec=loss_vs_n1(a, st, rank, world, dev, scaling, group1))      # noqa: E731
    state = {"tname": tname, "trec": trec, "want": want}

    def sharded(aa, sc, pre=None):
        """timed_sharded; if the peer exchange fails at one of its AGREED points (set-up exchange, peer_agree: every rank
        raises there together) the job drops it and runs again over torch.distributed's collectives."""
        err = None
        try:
            return timed_sharded(aa, rank, world, dev, sc, state["want"], pre_step=pre)
        except PeerFailed as e:
            err = str(e)
        except Exception as e:          # noqa: BLE001
            if state["tname"] != "peer" or "peer exchange set-up failed" not in str(e):
                raise
            err = str(e)
        gc.collect()
        torch.cuda.synchronize()
        mdist.disable_peer_exchange(None)
        state.update(tname="collective", want=False,
                     trec={"transport": "collective (%s), eager" % dist.get_backend(), "peer_rejected": err[:300]})
        return timed_sharded(aa, rank, world, dev, sc, False, pre_step=pre)
    r = sharded(a, scaling, check)
    tname, trec = state["tname"], state["trec"]
    stats = r["stats"]
    lrec = lbox.get("rec")
    devs = rank_devices(world, dev) if world > 1 else None
    out = result_line(a, world, scaling, r["ms"], stats["edge_layers_global"], stats, r["captured"], r["loss"],
                      "row-shard x%d, %s scheme, %d column chunk(s) per exchange (%s)" % (
                          world, stats["scheme"], stats["chunks"],
                          "peer push all-gather / pull reduce-scatter over IPC-mapped windows" if tname == "peer"
                          else "RCCL all-gather / reduce-scatter"),
                      stats["n_users"], stats["n_items"], stats["n_edges"])
    if rank == 0:
        out["roofline"] = spmm_roofline(r["plans"], r["mats"], a.d, traffic=False)   # rank 0's shard
        out["comm"] = r["comm"]
        out["comm"].update(trec)
        if devs is not None:
            out["rccl_ranks"] = devs
        if lrec is not None:
            out["loss_vs_n1"] = lrec
        full = committed_full_n1_figure()
        if a.workload == "synth" and world == 8 and full:      # this job IS configs[4]: strong ratio to the same graph on one GPU
            out["full_n1"] = full
            out["strong_vs_full_n1"] = round(full["ms_per_step"] / r["ms"], 3)
    # N > 1: the shape on which north_star's >= 6x is arithmetically possible, in the same run - configs[4]'s per-rank
    # share x N (250 K users x 125 K items x 12.5 M edges per rank, d = 128), a few steps, next to the committed one-rank
    # figure of the same share (its ratio = the weak-scaling efficiency on that shape)
    r2 = None
    if world > 1 and a.workload not in ("synth", "synth-full") and not a.no_stress:
        r = None
        gc.collect()
        torch.cuda.empty_cache()
        a2 = copy.copy(a)
        a2.workload, a2.d, a2.steps, a2.warmup = "synth", 128, min(a.steps, 10), min(a.warmup, 3)
        dog = stress_watchdog(a.stress_timeout, out, rank)
        try:
            r2 = sharded(a2, "weak")
            st2 = r2["stats"]
            rec = {"what": "configs[4]: per-rank share 250K users x 125K items x 12.5M edges, d=128, x %d ranks" % world,
                   "ms_per_step": round(r2["ms"], 4), "steps": a2.steps, "warmup": a2.warmup,
                   "edge_layers_per_s": round(st2["edge_layers_global"] / (r2["ms"] * 1e-3), 1),
                   "launch": "hipGraph replay" if r2["captured"] else "eager", "comm": r2["comm"],
                   "final_loss": round(r2["loss"], 6)}
            # the SAME share on ONE rank, measured in this invocation on rank 0 (no link crossed): the weak-scaling denominator
            try:
                rec["n1_ms"] = one_rank_ms(a2, rank, dev, group1)
                if rec["n1_ms"]:
                    rec["weak_efficiency_vs_n1_ms"] = round(rec["n1_ms"] / rec["ms_per_step"], 4)
            except Exception as e:
                rec["n1_ms_error"] = repr(e)[:300]
            ref = committed_rank_figure()
            if ref:           # WEAK: one rank's share alone (its item table 1/8 of the job's) vs the same share inside the job
                rec["one_rank"] = ref
                rec["per_rank_ratio"] = round(rec["edge_layers_per_s"] / world / ref["edge_layers_per_s"], 4)
                rec["speedup_vs_one_rank"] = round(rec["edge_layers_per_s"] / ref["edge_layers_per_s"], 3)
            full = committed_full_n1_figure()
            if full and world == 8:   # STRONG: the 8-rank job IS configs[4]; the same 2M x 1M x 100M graph on ONE GPU
                rec["full_n1"] = full
                rec["strong_vs_full_n1"] = round(full["ms_per_step"] / rec["ms_per_step"], 3)
        except Exception as e:       # the headline line must survive a failing stress run
            rec = {"error": repr(e)[:400]}
        dog.cancel()
        if rank == 0:
            out["scaling_stress"] = rec
    if rank == 0:
        print(json.dumps(out), flush=True)
    dist.barrier()
    r = r2 = None                     # (captured graphs before the process group: see the probe branch)
    gc.collect()
    torch.cuda.synchronize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
