"""Bytes a rank moves per hot-path step under row sharding, per xGMI link, against its compute time and the collectives'
launch latency - the arithmetic behind DESIGN.md section 6 (runs anywhere: python tools/shard_bytes.py [--link-gbs ...]).

Schemes for the 2 L GCN products and the packed modal chain of one step (every forward collective has an adjoint of the same
size in the backward):

  gather-both       (built, round 3; kept for A/B) all-gather the item table before A_ui . X_i AND the user table before
                    A_iu . X_u; adjoints = reduce-scatters of the same sizes
  item-side         (built, round 4, the default: dist._ShardedItemSide) a rank keeps only its users' edges; A_iu . X_u is a
                    local partial product over ALL items + a reduce-scatter of item-table size: every collective moves
                    item-table bytes. Per step: 2 L GCN-table passes (width d) + 2 modal passes (width nm d), twice
                    (forward + backward) = (4 L d + 4 nm d) x I x 4 bytes of full buffers
  item-side + repl  (built, round 5: ShardedMMSSL(replicate_feats=True)) the constant feature matrices live on every rank,
                    every rank projects ALL items: the projected features X never travel and their gradient is consumed as a
                    per-rank partial (the weight gradient is summed by the all-reduce the replicated parameters take
                    anyway): 2 of the 4 modal passes disappear, for (N - 1) x the projection flops per rank
  halo              (built, round 4) item-side moving only the item rows a rank's edges reference: x 0.98 of the bytes for
                    configs[4] (every rank touches nearly every item), x 0.56 for the Baby-shaped weak-scaling graph at N = 8
  2-D (R x C)       (not built) ranks in an R x C grid, A cut both ways: a gather inside a column group + a reduce-scatter
                    inside a row group per product

Model of a step at N ranks:  t = max(compute, link) + min(compute, link) / chunks + launches x latency
  compute  = measured one-GPU time of the rank's share (+ the replicated projection's extra flops)
  link     = bytes one rank receives / (N - 1) links / --link-gbs      (full mesh: one xGMI link per peer, traffic even)
  chunks   = column chunks per collective (chunk c's product under chunk c+1's transfer: only 1 / chunks of the shorter side
             stays exposed). Automatic: 1 below 64 MB per collective, 2 - 4 above (configs[4]: 4); the table prints one lane
             (no overlap) and four. launches x latency = collective launches per
             step x per-exchange cost: RCCL ~20 us per collective launch (measured on one rank, round 4); the PEER EXCHANGE
             (built, round 6: csrc/peer.hip - a push kernel + a wait kernel per all-gather, signal + wait + pull per
             reduce-scatter, all inside the step's hipGraph) ~8 us per exchange (2 - 3 graph nodes of ~3 us)
Speed-up = one-GPU time of the WHOLE problem / t. For configs[4] that denominator is measured: 139.7 ms per step for the whole
2M x 1M x 100M graph on one MI355X (profiles/r05_bench_synth_full_n1.json) - 8.0 x the 17.5 ms a rank's share costs."""
import argparse

ap = argparse.ArgumentParser()
ap.add_argument("--link-gbs", type=float, nargs="*", default=[45.0, 50.0, 60.0, 70.0, 75.0],
                help="achievable GB/s per xGMI link and direction (swept)")
ap.add_argument("--layers", type=int, default=3)
ap.add_argument("--latency-us", type=float, default=20.0, help="cost of one RCCL collective launch")
ap.add_argument("--peer-latency-us", type=float, default=10.0, help="fixed cost of one exchange over the peer transport (1 - 2 kernels; measured at world 1: profiles/r06/peer_world1.txt)")
ap.add_argument("--tflops", type=float, default=90.0, help="fp32 projection rate for the replicated-features extra work")
a = ap.parse_args()

N = 8


def step_ms(compute, link, chunks, launches):
    return max(compute, link) + min(compute, link) / chunks + launches * a.latency_us * 1e-3


def table(name, I, U, d, nm, feat_dims, one_gpu_ms, rank_ms, weak):
    print(name)
    L = a.layers
    frac = (N - 1) / N
    full = {   # bytes of the full (gathered / to-be-scattered) buffers of one step on one rank
        "gather-both": (2 * L * (I + U) * d + 2 * (I + U) * nm * d) * 4.0,
        "item-side": (4 * L * I * d + 4 * I * nm * d) * 4.0,
        "item-side + repl": (4 * L * I * d + 2 * I * nm * d) * 4.0,
    }
    R, C = 2, 4
    per = lambda rows_g, rows_rs: ((R - 1) / R * rows_g / C + (C - 1) / C * rows_rs / R)       # noqa: E731
    rows = lambda w: (per(I, U) + per(U, I)) * w * 4.0                                       # noqa: E731
    recv2d = 2 * (L * rows(d) + rows(nm * d))
    extra_ms = 2.0 * 2.0 * frac * I * sum(feat_dims) * d / (a.tflops * 1e12) * 1e3            # fwd + wgrad over all items
    print("  one GPU, whole problem: %.1f ms;  one rank's share alone: %.2f ms;  replicated projection: +%.2f ms per rank" % (
        one_gpu_ms, rank_ms, extra_ms))
    for scheme, by in full.items():
        recv = by * frac
        print("  %-17s full buffers %6.2f GB / step, received %6.2f GB / rank, %5.0f MB / link" % (
            scheme, by / 1e9, recv / 1e9, recv / (N - 1) / 1e6))
    print("  %-17s (not built)                     received %6.2f GB / rank, %5.0f MB / link" % (
        "2-D 2x4", recv2d / 1e9, recv2d / (N - 1) / 1e6))
    for nc, label in ((1, "ONE lane (--chunks 1: transfers and products one after the other)"),
                      (4, "4 column-chunk lanes (what the automatic rule picks for 512 MB tables)")):
        print("  speed-up over one GPU at N = 8 (%s), %s, per exchange: RCCL %.0f us | peer exchange %.0f us:" % (
            "weak: 8 x the rows" if weak else "strong: the same problem", label, a.latency_us, a.peer_latency_us))
        print("    %-9s" % "GB/s/link" + "".join("%22s" % s for s in ("item-side", "item-side + repl", "2-D 2x4 (not built)")))
        for gbs in a.link_gbs:
            cells = []
            for scheme, recv, comp, launches in (("item-side", full["item-side"] * frac, rank_ms, 14 * nc),
                                                 ("item-side + repl", full["item-side + repl"] * frac, rank_ms + extra_ms, 12 * nc),
                                                 ("2-D", recv2d, rank_ms, 28 * nc)):
                link = recv / (N - 1) / (gbs * 1e9) * 1e3
                t = (comp + link if nc == 1 else max(comp, link) + min(comp, link) / nc) + launches * a.latency_us * 1e-3
                lat = launches * a.latency_us * 1e-3
                t3 = t - lat + launches * a.peer_latency_us * 1e-3
                tot = one_gpu_ms * (N if weak else 1)
                cells.append("%5.1fx | %4.1fx (%4.1f ms)" % (tot / t, tot / t3, t))
            print("    %-9.0f" % gbs + "".join("%22s" % c for c in cells))
    print()


# configs[4]: one-GPU whole-problem step and the per-rank share (1/8 of it: the SpMM time is linear in the edges)
table("configs[4]: 2M users x 1M items, 100M edges, d = 128, two 128-wide features (strong scaling, N = 8 is the config)",
      I=1_000_000, U=2_000_000, d=128, nm=2, feat_dims=(128, 128), one_gpu_ms=139.7, rank_ms=139.7 / 8, weak=False)
# Baby x 8 (weak scaling: the driver's --gpus 8 default): a 0.49 ms step against >= 14 collective launches
table("Amazon-Baby shape x 8 (weak scaling: every rank one Baby-sized share), d = 64, 4096 + 1024 wide features",
      I=18357 * 8, U=35598 * 8, d=64, nm=2, feat_dims=(4096, 1024), one_gpu_ms=0.49, rank_ms=0.49, weak=True)
