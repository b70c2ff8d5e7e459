"""Bytes a rank receives per hot-path step under row sharding, per xGMI link, against its compute time - the arithmetic
behind DESIGN.md section 6 (runs anywhere: python tools/shard_bytes.py). Three schemes for the 2L GCN products and the
packed modal chain of one step (forward gathers + backward reduce-scatters move the same bytes):

  gather-both      (built, round 3: mmssl_amd/dist.py _ShardedHotForward, bench.py --scheme gather-both) all-gather the item
                   table before A_ui . X_i AND the user table before A_iu . X_u; the adjoints are reduce-scatters of the
                   same sizes
  item-collectives (built, round 4, the default: _ShardedItemSide, --scheme item-side) every rank keeps only its user-row
                   block of the graph; A_iu . X_u becomes local partial products over ALL items + a reduce-scatter of
                   item-table size (its adjoint a gather of item-table size): every collective moves item-table bytes; with
                   --chunks c every collective and the products around it are cut into c column chunks on c lanes, so
                   that a chunk's product runs under the next chunk's transfer
  halo             (built, round 4: --scheme halo) item-collectives moving only the item rows a rank's edges
                   reference: x 0.92 / 0.76 / 0.56 of the bytes for the Baby-shaped weak-scaling graph at N = 2 / 4 / 8 (measured
                   fractions of referenced rows), x 0.98 for configs[4]
  2-D (R x C)      (not built) ranks in an R x C grid, A cut in both directions: a gather inside a column group and a
                   reduce-scatter inside a row group per product

Per-link time assumes one xGMI link per peer (full mesh, 8 GPUs) at `--link-gbs` per direction and that a rank's traffic
spreads evenly over its N - 1 peers."""
import argparse

ap = argparse.ArgumentParser()
ap.add_argument("--link-gbs", type=float, default=70.0, help="achievable GB/s per xGMI link and direction")
ap.add_argument("--layers", type=int, default=3)
a = ap.parse_args()

SHAPES = {   # users, items, d, modal width, per-rank compute ms of one step (measured on one MI355X), weak: rows grow with N
    "baby x N (weak)": dict(U=35598, I=18357, d=64, dm=128, ms=0.54, weak=True),
    "configs[4] 2M x 1M (fixed, N = 8 is the config)": dict(U=2_000_000, I=1_000_000, d=128, dm=256, ms=16.4, weak=False),
}


def passes(U, I, d, dm, L):
    """(bytes of item-table-sized passes, bytes of user-table-sized passes) per step, forward + backward."""
    item = 2 * (L * I * d + I * dm) * 4        # gathers of items (fwd) + their reduce-scatters (bwd)
    user = 2 * (L * U * d + U * dm) * 4
    return item, user


for name, s in SHAPES.items():
    print(name)
    for N in (2, 4, 8):
        k = N if s["weak"] else 1
        U, I = s["U"] * k, s["I"] * k
        item, user = passes(U, I, s["d"], s["dm"], a.layers)
        frac = (N - 1) / N
        schemes = {"gather-both": (item + user) * frac, "item-collectives": 2 * item * frac}
        if N == 8:
            R, C = 2, 4                       # rows of A cut R ways (user side), columns C ways (item side)
            # A_ui product: gather X_i inside a column group (R ranks share a column band), reduce-scatter partial user rows
            # across the C ranks of a row group; A_iu product symmetric
            per = lambda rows_g, rows_rs: ((R - 1) / R * rows_g / C + (C - 1) / C * rows_rs / R)       # noqa: E731
            rows = lambda w: (per(I, U) + per(U, I)) * w * 4                                         # noqa: E731
            schemes["2-D 2x4"] = 2 * (a.layers * rows(s["d"]) + rows(s["dm"]))
        line = []
        for nm, b in schemes.items():
            per_link = b / (N - 1)
            ms = per_link / (a.link_gbs * 1e9) * 1e3
            line.append("%s %.0f MB/rank = %.0f MB/link = %.2f ms" % (nm, b / 1e6, per_link / 1e6, ms))
        comp = s["ms"] if s["weak"] else s["ms"] * 8 / N
        print("  N=%d  compute %.2f ms |  %s" % (N, comp, "  |  ".join(line)))
        # what the built item-side scheme can reach: no overlap (links idle while the SpMMs run) .. perfect overlap with c
        # column chunks (only the first chunk's transfer and the last chunk's product are exposed per collective)
        link = schemes["item-collectives"] / (N - 1) / (a.link_gbs * 1e9) * 1e3
        one = s["ms"] if s["weak"] else s["ms"] * 8
        for c in (1, 2, 4):
            t = max(comp, link) + (min(comp, link) / c if c > 1 else min(comp, link))
            print("         item-side, %d chunk(s): %.2f ms per step -> %.1fx one GPU%s" % (
                c, t, (one / t) if not s["weak"] else N * one / t / 1.0,
                "" if c > 1 else "  (no overlap)"))
