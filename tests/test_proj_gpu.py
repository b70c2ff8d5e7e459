"""Grouped projection kernels (csrc/projection.hip: all modalities of a step in one stream-K launch) against torch-CPU
fp32 of the same op: nn.Linear + nn.Dropout of every modality (Models.py:28-29, 54, 173-174) and its weight / bias
gradients. Shapes: Baby (configs[2]), Tiktok V/A/T (configs[1]), ragged row counts, a single problem."""
import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(params=["split", "f32"])
def precision(request, monkeypatch):
    """A test that asks for this fixture runs twice: on the split-precision kernels (the default: exact 3-way bf16 cut, six partial
    products on the bf16 matrix pipe) and on the fp32-MFMA kernels (ops.PROJ_SPLIT = False)."""
    from mmssl_amd import ops
    monkeypatch.setattr(ops, "PROJ_SPLIT", request.param == "split")
    return request.param


def _problem(M, Ks, seed):
    g = torch.Generator().manual_seed(seed)
    Fs = [torch.randn(M, k, generator=g) for k in Ks]
    Ws = [torch.randn(64, k, generator=g) * (1.0 / k ** 0.5) for k in Ks]
    bs = [torch.randn(64, generator=g) * 0.1 for _ in Ks]
    keep = (torch.rand(len(Ks), M, 64, generator=g) >= 0.2).to(torch.uint8)
    return Fs, Ws, bs, keep


@pytest.mark.parametrize("M,Ks", [(18357, (4096, 1024)), (6710, (128, 768, 128)), (1000, (64,)), (257, (32, 96)),
                                  (256, (4096,))])
def test_proj_forward_matches_torch(M, Ks, precision):
    from mmssl_amd import ops
    assert ops.proj_supported(Ks, M, 64)
    Fs, Ws, bs, keep = _problem(M, Ks, 1)
    Fd, Wd, bd = [f.to(DEV) for f in Fs], [w.to(DEV) for w in Ws], [b.to(DEV) for b in bs]
    scale = 1.25
    Y, _ = ops.proj_forward(Fd, Wd, bd, keep=keep.to(DEV), scale=scale)
    assert Y.shape == (M, 64 * len(Ks))
    Y2, _ = ops.proj_forward(Fd, Wd, bd, keep=keep.to(DEV), scale=scale)
    assert torch.equal(Y, Y2)                                   # fixed-order partial sums: bitwise reproducible
    Yn, _ = ops.proj_forward(Fd, Wd, [None] * len(Ks), scale=1.0)       # eval mode: no bias here, no dropout
    for g in range(len(Ks)):
        lin = Fs[g] @ Ws[g].t()
        ref = (lin + bs[g]) * keep[g].float() * scale
        assert H.rel_err(Y[:, 64 * g:64 * g + 64].cpu(), ref) < 2e-5, (g, Ks[g])
        assert H.rel_err(Yn[:, 64 * g:64 * g + 64].cpu(), lin) < 2e-5, (g, Ks[g])
        # the same numbers as the per-modality kernel of ops.linear (same fp32 MFMA arithmetic, other summation order)
        one = ops.linear(Fd[g], Wd[g], bd[g], keep[g].to(DEV), scale)
        assert H.rel_err(Y[:, 64 * g:64 * g + 64].cpu(), one.cpu()) < 2e-5


def test_proj_forward_draws_the_masks_of_dropout_masks(precision):
    """draw=(p, state): the epilogue's inline generator reproduces ops.dropout_masks at the same generator state, the
    output is the given-mask result for those masks, and the state itself is left to the caller (external tick)."""
    from mmssl_amd import ops
    M, Ks = 5000, (96, 160)
    Fs, Ws, bs, _ = _problem(M, Ks, 2)
    Fd, Wd, bd = [f.to(DEV) for f in Fs], [w.to(DEV) for w in Ws], [b.to(DEV) for b in bs]
    dev = torch.device(DEV, torch.cuda.current_device())
    ops.seed_dropout(77, dev)
    st = ops._rng_state(dev)
    before = st.clone()
    Y, keep = ops.proj_forward(Fd, Wd, bd, draw=(0.2, st), scale=1.25)
    assert torch.equal(st, before)
    want = ops.dropout_masks(len(Ks), M, 64, 0.2, dev)          # same state -> same bytes; advances the counter
    assert torch.equal(keep, want)
    assert 0.78 < float(keep.float().mean()) < 0.82
    Yg, _ = ops.proj_forward(Fd, Wd, bd, keep=want.contiguous(), scale=1.25)
    assert torch.equal(Y, Yg)
    _, keep2 = ops.proj_forward(Fd, Wd, bd, draw=(0.2, st), scale=1.25)     # next counter value: other masks
    assert not torch.equal(keep2, keep)


@pytest.mark.parametrize("M,Ks", [(18357, (4096, 1024)), (6710, (128, 768, 128)), (1000, (64,)), (257, (32, 96)),
                                  (33, (260,))])
def test_proj_wgrad_matches_torch(M, Ks, precision):
    from mmssl_amd import ops
    assert ops.proj_supported(Ks, M, 64, wgrad=True)
    Fs, _, _, _ = _problem(M, Ks, 3)
    g = torch.Generator().manual_seed(9)
    G = torch.randn(M, 64 * len(Ks), generator=g)
    G[G.abs() < 0.25] = 0.0                                       # like a dropout-masked gradient
    Fd, Gd = [f.to(DEV) for f in Fs], G.to(DEV)
    gW, gb = ops.proj_wgrad(Gd, Fd)
    gW2, gb2 = ops.proj_wgrad(Gd, Fd)
    for k in range(len(Ks)):
        Gk = G[:, 64 * k:64 * k + 64]
        assert H.rel_err(gW[k].cpu(), Gk.t() @ Fs[k]) < 2e-5, (k, Ks[k])
        assert H.rel_err(gb[k].cpu(), Gk.sum(0)) < 2e-5, (k, Ks[k])
        assert torch.equal(gW[k], gW2[k]) and torch.equal(gb[k], gb2[k])
    # a G that is a column slice of a wider buffer (row pitch > 64 n)
    wide = torch.zeros(M, 64 * len(Ks) + 64, device=DEV)
    wide[:, :64 * len(Ks)] = Gd
    gW3, _ = ops.proj_wgrad(wide[:, :64 * len(Ks)], Fd)
    assert all(torch.equal(a, b) for a, b in zip(gW, gW3))


def test_proj_rejects_what_it_cannot_run(precision):
    from mmssl_amd import ops, _lib
    bad_k = 102 if precision == "split" else 100                 # split: K % 4; fp32-MFMA forward: K % 32
    assert not ops.proj_supported((bad_k,), 1000, 64)
    assert not ops.proj_supported((128,), 1000, 32)               # N != 64
    assert ops.proj_supported((100,), 1000, 64, wgrad=True)
    assert ops.proj_supported((100,), 1000, 64) == (precision == "split")
    F_ = torch.randn(100, bad_k, device=DEV)
    with pytest.raises(_lib.MmsslError):
        ops.proj_forward([F_], [torch.randn(64, bad_k, device=DEV)], [None])


@pytest.mark.parametrize("M,Ks", [(18357, (4096, 1024)), (3000, (20, 260))])
def test_split_precision_is_fp32_accurate_against_float64(M, Ks):
    """The claim behind the split-precision kernels: cutting every fp32 value exactly into three bf16 pieces and keeping
    the six partial products of weight >= 2^-16 loses at most ~2^-23 of a product - one fp32 rounding. Measured against a
    float64 product, element by element (normalised by |F| . |W|, the scale of a dot product's rounding error), the error
    of this path must not exceed that of fp32 arithmetic itself: 1.25 x the fp32-MFMA kernels' (same sequential fp32
    accumulation; a three-product bf16x3 scheme would sit at 3 x), within 4 x torch's blocked fp32 GEMM on the CPU. Forward and weight gradient; K = 20 exercises the zero padding of
    a slice, M = 18357 the ragged last tile and reduction slice."""
    from mmssl_amd import ops
    assert ops.PROJ_SPLIT
    Fs, Ws, bs, _ = _problem(M, Ks, 11)
    Fd, Wd = [f.to(DEV) for f in Fs], [w.to(DEV) for w in Ws]
    Y, _ = ops.proj_forward(Fd, Wd, [None] * len(Ks), scale=1.0)
    ops.PROJ_SPLIT = False
    try:
        Yf = ops.proj_forward(Fd, Wd, [None] * len(Ks), scale=1.0)[0] if all(k % 32 == 0 for k in Ks) else None
    finally:
        ops.PROJ_SPLIT = True
    g = torch.Generator().manual_seed(5)
    G = torch.randn(M, 64 * len(Ks), generator=g)
    gW, _ = ops.proj_wgrad(G.to(DEV), Fd)
    for k in range(len(Ks)):
        F64, W64 = Fs[k].double(), Ws[k].double()
        ref = F64 @ W64.t()
        scale = F64.abs() @ W64.abs().t()                          # sum |a| |b|: what a dot product's rounding scales with
        e_split = float(((Y[:, 64 * k:64 * k + 64].cpu().double() - ref).abs() / scale).max())
        e_torch = float((((Fs[k] @ Ws[k].t()).double() - ref).abs() / scale).max())
        assert e_split <= 4.0 * e_torch + 1e-9 and e_split < 1e-6, (Ks[k], e_split, e_torch)
        if Yf is not None:
            e_mfma = float(((Yf[:, 64 * k:64 * k + 64].cpu().double() - ref).abs() / scale).max())
            assert e_split <= 1.25 * e_mfma + 1e-9, (Ks[k], e_split, e_mfma)
        Gk = G[:, 64 * k:64 * k + 64].double()
        refw = Gk.t() @ F64
        scw = Gk.abs().t() @ F64.abs()
        e_w = float(((gW[k].cpu().double() - refw).abs() / scw).max())
        e_wt = float((((G[:, 64 * k:64 * k + 64].t() @ Fs[k]).double() - refw).abs() / scw).max())
        assert e_w <= 4.0 * e_wt + 1e-9 and e_w < 1e-6, (Ks[k], e_w, e_wt)


def test_split_precision_images_follow_the_feature_matrix():
    """The packed images are cached per feature matrix: a second matrix of the same shape gets its own images (the cache
    keeps the first alive, so its address cannot be recycled), an in-place change rebuilds them."""
    from mmssl_amd import ops
    assert ops.PROJ_SPLIT
    M, K = 700, 96
    W = [torch.randn(64, K, device=DEV) * 0.1]
    F1 = torch.randn(M, K, device=DEV)
    Y1, _ = ops.proj_forward([F1], W, [None])
    F2 = torch.randn(M, K, device=DEV)
    Y2, _ = ops.proj_forward([F2], W, [None])
    assert H.rel_err(Y1.cpu(), (F1 @ W[0].t()).cpu()) < 2e-5 and H.rel_err(Y2.cpu(), (F2 @ W[0].t()).cpu()) < 2e-5
    F1.mul_(2.0)
    Y3, _ = ops.proj_forward([F1], W, [None])
    assert H.rel_err(Y3.cpu(), 2 * Y1.cpu()) < 1e-6
    for _ in range(12):                                            # more matrices than cache entries: evictions
        Fn = torch.randn(M, K, device=DEV)
        Yn, _ = ops.proj_forward([Fn], W, [None])
        assert H.rel_err(Yn.cpu(), (Fn @ W[0].t()).cpu()) < 2e-5


def test_per_modality_weight_gradient_is_bit_stable_beside_memory_bound_streams():
    """csrc/linear.hip wgrad10_kernel at configs[4]'s shape ([1M, 128]^T x [1M, 128], with and without the fused dropout
    backward) while three other streams saturate the memory system with HBM-resident transposed SpMMs and a fourth with
    device copies: 30 launches of each kind, every result the same bits as on the idle device. The round-5 kernel
    (registers copied while their loads were in flight) fails this when its loads are late enough:
    tools/wgrad_race_repro.py is the A/B of the two builds, profiles/r06/wgrad_race_repro.txt its record."""
    from mmssl_amd import ops, synth
    from mmssl_amd.graph import GraphPlan
    dev = torch.device("cuda")
    raw = synth.interaction_matrix(250_000, 1_000_000, 12_500_000, seed=1000, item_seed=77)
    plan = GraphPlan(synth.normalised_rows(raw), xcd_bands=-1)
    g = torch.Generator().manual_seed(0)
    rows = 1_000_000
    G = torch.randn(rows, 128, generator=g).to(dev)
    F_ = torch.randn(rows, 128, generator=g).to(dev)
    keep = (torch.rand(rows, 128, generator=g) >= 0.2).to(torch.uint8).to(dev)
    W = torch.empty(128, 128, device=dev)
    Gu = [torch.randn(250_000, 128, generator=g).to(dev) for _ in range(3)]
    big = [torch.empty(256 << 20, dtype=torch.uint8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    ref = {"plain": ops._linear_wgrad_raw(G, None, 1.0, F_, W)[1].clone(),
           "masked": ops._linear_wgrad_raw(G, keep, 1.25, F_, W)[1].clone()}
    # against float64 on a slice of the reduction (the whole product on the CPU would take a minute)
    sub = slice(0, 200_000)
    want = (G[sub].double().t() @ F_[sub].double()).float()
    got = ops._linear_wgrad_raw(G[sub].contiguous(), None, 1.0, F_[sub].contiguous(), W)[1]
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    main_s, copy_s = torch.cuda.Stream(), torch.cuda.Stream()
    hog_s = [torch.cuda.Stream() for _ in range(3)]
    for it in range(10):
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
                assert torch.equal(o, ref[k]), (it, k, float((o - ref[k]).abs().max()), float(ref[k].abs().max()))


def test_grouped_projection_is_bit_stable_beside_memory_bound_streams():
    """The same question for the hot path's split-precision projection at the Baby shape (csrc/projection.hip
    projx_sk_kernel keeps four slices of its long operand in flight in registers): forward and weight gradient, 30
    launches each under the same load, the bits of the idle device every time."""
    from mmssl_amd import ops, synth
    from mmssl_amd.graph import GraphPlan
    dev = torch.device("cuda")
    raw = synth.interaction_matrix(250_000, 1_000_000, 12_500_000, seed=1000, item_seed=77)
    plan = GraphPlan(synth.normalised_rows(raw), xcd_bands=-1)
    M, Ks = 18357, (4096, 1024)
    Fs, Ws, bs, keep = _problem(M, Ks, 11)
    Fd, Wd, bd, keep = [f.to(dev) for f in Fs], [w.to(dev) for w in Ws], [b.to(dev) for b in bs], keep.to(dev)
    g = torch.Generator().manual_seed(1)
    Gd = torch.randn(M, 64 * len(Ks), generator=g).to(dev)
    Gu = [torch.randn(250_000, 128, generator=g).to(dev) for _ in range(3)]
    big = [torch.empty(256 << 20, dtype=torch.uint8, device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    Y0 = ops.proj_forward(Fd, Wd, bd, keep=keep, scale=1.25)[0].clone()
    gW0, gb0 = ops.proj_wgrad(Gd, Fd)
    gW0, gb0 = [t.clone() for t in gW0], [t.clone() for t in gb0]
    torch.cuda.synchronize()
    main_s, copy_s = torch.cuda.Stream(), torch.cuda.Stream()
    hog_s = [torch.cuda.Stream() for _ in range(3)]
    for it in range(10):
        for h, st in enumerate(hog_s):
            with torch.cuda.stream(st):
                for _ in range(3):
                    ops._spmm_raw(plan.twin(h + 1), True, Gu[h], ops.EPI_NONE)
        with torch.cuda.stream(copy_s):
            for _ in range(4):
                big[1].copy_(big[0])
        with torch.cuda.stream(main_s):
            Ys = [ops.proj_forward(Fd, Wd, bd, keep=keep, scale=1.25)[0] for _ in range(3)]
            gs = [ops.proj_wgrad(Gd, Fd) for _ in range(3)]
        torch.cuda.synchronize()
        for Y in Ys:
            assert torch.equal(Y, Y0), (it, "forward", float((Y - Y0).abs().max()))
        for gW, gb in gs:
            for a, b in zip(list(gW) + list(gb), gW0 + gb0):
                assert torch.equal(a, b), (it, "wgrad", float((a - b).abs().max()), float(b.abs().max()))

