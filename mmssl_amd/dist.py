"""Row-sharded multi-GPU hot path (one process per GPU; the exchanges below go over torch.distributed / RCCL or - after
enable_peer_exchange(), bench.py's default for N > 1 - through kernels writing and reading IPC-mapped peer windows:
mmssl_amd/peer.py, csrc/peer.hip; the same call sites, the same data movement).

The reference is single-process / single-GPU (no torch.distributed anywhere: SURVEY.md 2.2), so
this is a new design whose correctness bar is "N-rank result == 1-rank result == reference".

Partition: users and items are cut into `world` contiguous, equally sized row
blocks (padded with empty rows to a multiple of `world`). Rank r owns
    E_user[U_r], E_item[I_r], the raw feature rows of I_r,
    the CSR rows A_ui[U_r, :] and A_iu[I_r, :] (column indices stay global).
Exchange: before every Y_r = A_r . X the row shards of X are ALL-GATHERed; the backward of that
gather is a REDUCE-SCATTER of the partial A_r^T . gY_r. Row-local work (projection of the local
item rows, normalise, softmax epilogue, layer mean, fusion) needs no communication. The batch
rows for BPR / InfoNCE are assembled by an all-reduce of zero-padded [B, d] buffers (each row has
exactly one non-zero contributor, so the sum is exact) and the losses are evaluated replicated;
each rank back-propagates only into the rows it owns. Replicated dense parameters (projection
weights, fusion weights) get their gradients all-reduced in one flat bucket.

xGMI note: RCCL ring collectives are bound by one ~153 GB/s link; the per-layer gather is
N x 4.7 MB for the Baby shape (latency-bound) and 0.5-1 GB for the 100M-edge stress shape, where
overlapping the gather with the local-column part of the SpMM is the lever.

The compute kernels are reached through a small backend object so that the sharding logic can
be exercised on CPU with `gloo` (tests/test_dist_cpu.py plugs the oracle in); the product
backend is `HipBackend` (libmmssl_hip.so) and there is no CPU fallback in this package.
"""
import os

import numpy as np
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.nn as nn


class HipBackend:
    """The product compute backend: HIP kernels via mmssl_amd.ops. One backend object per sharded model / step: it owns
    the node's side streams and the hand-offs between a step's forward, loss section and backward (the regulariser
    partials, the event behind the backward's fuse kernel) - the sharded counterpart of hotnode.HotCtx."""

    def __init__(self):
        from . import ops
        from .graph import GraphPlan
        self.ops = ops
        self.make_graph = GraphPlan
        self.EPI_NONE, self.EPI_SOFTMAX = ops.EPI_NONE, ops.EPI_SOFTMAX
        for name in ("spmm", "l2norm_rows", "linear", "bpr", "infonce", "sumsq"):
            setattr(self, name, getattr(ops, name))
        self.softmax_rows = ops.softmax_rows_fn
        # non-autograd kernels for the fused sharded node
        self.spmm_raw = ops._spmm_raw
        self.softmax_rows_bwd = ops.softmax_rows_bwd
        self.EPI_AXPY, self.EPI_AXPY_SOFTMAX_BWD = ops.EPI_AXPY, ops.EPI_AXPY_SOFTMAX_BWD
        self.dropout_masks = ops.dropout_masks
        self.loss_assemble = ops.loss_assemble
        self._ar = {}
        self._streams = {}
        self.after_fuse_bwd = None      # event of the latest packed node's backward (see _ShardedHotForward.backward)
        self.tables_stream = None       # the stream on which that backward completes the gradient of i_0
        self.table_grads = None
        self.defer_ss, self.ss_parts = False, None      # hand-off of the regulariser partials to a step's loss tail
        # batch-rows form (set by a step object for one forward + backward): (global user rows [B], global item rows [2B],
        # lo_u, lo_i) - the ONLY rows of the fused tables the step's loss reads. The forward fuses the owned ones of those
        # rows only; the regulariser's |Mod|^2 sums then come out of the backward's fuse kernel (reg_parts = the partials,
        # reg_ss = the 0-dim tensor that receives their sum when the step adds c * sum to its loss)
        self.batch_rows, self.reg_parts, self.reg_ss = None, None, None

    def gather_owned(self, table, idx, lo, out):
        from . import _lib
        rc = _lib.lib().mmssl_gather_owned_rows_f32(table.data_ptr(), table.shape[0], table.shape[1], idx.data_ptr(),
                                                    idx.shape[0], int(lo), out.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "mmssl_gather_owned_rows_f32")

    def pack_rows(self, table, idx):
        """table[idx] (idx int64 device rows of `table`): the halo exchange's send buffer, one gather launch."""
        out = torch.empty((idx.shape[0], table.shape[1]), dtype=table.dtype, device=table.device)
        if idx.shape[0]:
            self.gather_owned(table.contiguous(), idx, 0, out)
        return out

    def scatter_owned(self, g, idx, lo, gtable):
        from . import _lib
        rc = _lib.lib().mmssl_scatter_owned_rows_f32(g.data_ptr(), idx.data_ptr(), idx.shape[0], int(lo), gtable.shape[0],
                                                     gtable.shape[1], gtable.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "mmssl_scatter_owned_rows_f32")

    def side_streams(self, device):
        """Three side streams for the sharded node's chains, owned by this backend object."""
        key = (device.type, device.index)
        st = self._streams.get(key)
        if st is None:
            st = self._streams[key] = [torch.cuda.Stream(device=device) for _ in range(3)]
        return st

    def lane_streams(self, device, n):
        """Streams for the column-chunk lanes of the item-side node (grown on demand, owned by this backend object)."""
        key = ("lanes", device.type, device.index)
        st = self._streams.setdefault(key, [])
        while len(st) < n:
            st.append(torch.cuda.Stream(device=device))
        return st[:n]

    def softmax_rows_(self, X):
        return self.ops.softmax_rows(X, out=X)

    @staticmethod
    def chunk_ok(width, nc):
        return nc >= 1 and width % nc == 0 and (width // nc) in (32, 64, 128, 256)

    def _identity(self, B, dev):
        ar = self._ar.get((B, dev))
        if ar is None:
            base = torch.arange(B, dtype=torch.int64, device=dev)
            ar = (base, base + B)
            self._ar[(B, dev)] = ar
        return ar

    def batch_losses_rows(self, u, ia, z_img, z_txt, decay, batch_size, tau, eager_w=None, tail=None, hot=None):
        """[mf, emb, 0, cl_img, cl_txt] from already gathered rows (u [B, d]; ia [2B, d] = positive then negative items):
        ONE fused node (BPR + both InfoNCE problems, see ops._BatchLosses) fed with identity indices. eager_w + tail: the
        single-chain form whose last launch also assembles the loss and ticks the step's counters (ops.batch_losses_vec)."""
        ar = self._identity(u.shape[0], u.device)
        return self.ops.batch_losses_vec(u, ia, z_img, z_txt, ar[0], ar[0], ar[1], decay, batch_size, tau,
                                         hot=hot, eager_w=eager_w, tail=tail)

    # ---- the packed node's kernels (the grouped projection and the two-sided fuse kernels of the unsharded hot node) ----
    def packed_supported(self, feat_dims, rows, d):
        """The packed node runs every modality list whose packed width nm * d the SpMM / fuse kernels have; the projection
        runs grouped (csrc/projection.hip: d == 64, feature widths in whole 32-deep slices) or per modality
        (csrc/linear.hip: any width % 4, e.g. configs[4]'s d = 128)."""
        nm = len(feat_dims)
        return (1 <= nm <= 4 and (nm * d) in (32, 64, 128, 256) and d in (32, 64, 128, 256) and (d // 4) * nm <= 64
                and all(int(k) % 4 == 0 for k in feat_dims))

    def _grouped(self, Ks, M, d):
        return self.ops.proj_supported(Ks, M, d) and self.ops.proj_supported(Ks, M, d, wgrad=True)

    def proj_forward(self, Fs, Ws, bs, keep, scale, draw_p=0.0, external_tick=False):
        """(X [M, nm d], keep). keep None and draw_p > 0: fresh masks from the device generator (drawn in the grouped
        projection's epilogue, or by one mask launch in front of the per-modality kernels)."""
        ops = self.ops
        M, d = Fs[0].shape[0], Ws[0].shape[0]
        dev = Fs[0].device
        if self._grouped([f.shape[1] for f in Fs], M, d):
            if keep is None and draw_p > 0.0:
                X, keep = ops.proj_forward(Fs, Ws, bs, draw=(draw_p, ops._rng_state(dev)), scale=scale,
                                           blocks=ops.proj_step_blocks(dev))
                if not external_tick:
                    ops.tick_rng(dev)
                return X, keep
            return ops.proj_forward(Fs, Ws, bs, keep=keep, scale=scale, blocks=ops.proj_step_blocks(Fs[0].device))[0], keep
        if keep is None and draw_p > 0.0:
            keep = ops.dropout_masks(len(Fs), M, d, draw_p, dev, **({"external_tick": True} if external_tick else {}))
        cols = [ops._linear_raw(F_, W, b, None if keep is None else keep[m], scale)
                for m, (F_, W, b) in enumerate(zip(Fs, Ws, bs))]
        return torch.cat(cols, 1), keep

    def proj_wgrad(self, G, Fs, want_bias):
        """([gW_m], [gb_m] or None) from the already masked packed gradient G [M, nm d]."""
        ops = self.ops
        nm = len(Fs)
        M, d = Fs[0].shape[0], G.shape[1] // nm
        if self._grouped([f.shape[1] for f in Fs], M, d):
            return ops.proj_wgrad(G, Fs, want_bias=want_bias, blocks=ops.proj_step_blocks(G.device))
        gWs, gbs = [], []
        for m, F_ in enumerate(Fs):
            Gm = G[:, m * d:(m + 1) * d].contiguous()
            W_like = torch.empty((d, F_.shape[1]), dtype=torch.float32, device=G.device)
            _, gW, gb = ops._linear_wgrad_raw(Gm, None, 1.0, F_, W_like)
            gWs.append(gW)
            gbs.append(gb)
        return gWs, (gbs if want_bias else None)

    def fuse_fwd(self, us, MU, its, MI, inv, nm, r):
        """(u_g, i_g, ss): both sides in one launch; ss = |MU|^2 + |MI|^2 over the local rows (0-dim tensor)."""
        from . import _lib
        ops = self.ops
        d = us[0].shape[1]
        if self.batch_rows is not None:
            ru, ri, lo_u, lo_i = self.batch_rows
            u_g, i_g = ops.fuse_fwd_rows([(us, MU, ru), (its, MI, ri)], inv, nm, r, lo=(lo_u, lo_i))
            # UNDEFINED until the step has summed the backward fuse kernel's partials into it (ShardedHotPathStep.backward
            # raises if they never came); not pre-filled: that would be one more launch in every step
            ss = torch.empty((), dtype=torch.float32, device=MU.device)
            self.reg_ss, self.reg_parts = ss, None          # the backward's fuse kernel owes the |Mod|^2 partials
            return u_g, i_g, ss
        nbu, nbi = ops.fuse_blocks(us[0].shape[0], d, nm), ops.fuse_blocks(its[0].shape[0], d, nm)
        part = torch.empty(nbu + nbi, dtype=torch.float32, device=MU.device)
        u_g, i_g = ops.fuse_fwd([(us, MU, part[:nbu]), (its, MI, part[nbu:])], inv, nm, r)
        ss = torch.empty((), dtype=torch.float32, device=MU.device)
        if self.defer_ss:            # the step's loss tail reduces the partials (and stores the sum into ss): one launch less
            self.ss_parts = (ss, part)
        else:
            _lib.check(_lib.lib().mmssl_sum_partials_f32(part.data_ptr(), nbu + nbi, ss.data_ptr(), _lib.stream_ptr()),
                       "mmssl_sum_partials_f32")
        return u_g, i_g, ss

    def fuse_bwd(self, MU, Gu, G_MU, MI, Gi, G_MI, nm, r, inv, g_ss):
        """(gMU, g_u0, gMI): normalise backward + regulariser gradient of both sides in one launch."""
        part = None
        if self.reg_ss is not None and self.reg_parts is None:      # a batch-rows forward: leave the regulariser's partials
            ops = self.ops
            d = Gu.shape[1]
            nbu, nbi = ops.fuse_blocks(MU.shape[0], d, nm), ops.fuse_blocks(MI.shape[0], d, nm)
            part = torch.empty(nbu + nbi, dtype=torch.float32, device=MU.device)
            self.reg_parts = part
        (gMU, g_u0), (gMI, _) = self.ops.fuse_bwd([(MU, Gu, G_MU, True), (MI, Gi, G_MI, False)], nm, r, inv, g_ss, 2.0,
                                                  sumsq_part=None if part is None else [part[:nbu], part[nbu:]])
        return gMU, g_u0, gMI

    def spmm_mask(self, plan, X, keep, dm, scale):
        return self.ops.spmm_mask_raw(plan, True, X, keep, dm, scale)

    def mask_packed(self, G, keep, dm, scale):
        return self.ops.mask_packed(G, keep, dm, scale)


# ---------------------------------------------------------------------------------------------
# partition helpers
# ---------------------------------------------------------------------------------------------
class RowShard:
    def __init__(self, n, world, rank):
        self.n = int(n)
        self.per = -(-self.n // world)
        self.n_pad = self.per * world
        self.lo = rank * self.per
        self.hi = self.lo + self.per

    def slice_rows(self, t):
        """Rows [lo, hi) of a global [n, ...] tensor/array, zero-padded past n."""
        out_shape = (self.per,) + tuple(t.shape[1:])
        if isinstance(t, np.ndarray):
            out = np.zeros(out_shape, t.dtype)
        else:
            out = torch.zeros(out_shape, dtype=t.dtype)
        k = max(0, min(self.hi, self.n) - self.lo)
        if k > 0:
            out[:k] = t[self.lo:self.lo + k]
        return out


def shard_graph(mat, row_shard, col_shard):
    """Local CSR rows [lo, hi) of a global scipy matrix, columns global, padded to
    [per, n_cols_pad]."""
    m = sp.csr_matrix(mat, dtype=np.float32).copy()
    m.resize((row_shard.n_pad, col_shard.n_pad))
    return m[row_shard.lo:row_shard.hi].tocsr()


def shard_graph_cols(mat, row_shard, col_shard):
    """Local COLUMNS [lo, hi) of a global scipy matrix over ALL (padded) rows: [n_rows_pad, per]. The item-side scheme's
    second graph: A_iu[:, U_r] - the edges of this rank's users seen from the item side, so that A_iu . X_u becomes a
    local partial product over all items + a reduce-scatter."""
    m = sp.csr_matrix(mat, dtype=np.float32).copy()
    m.resize((row_shard.n_pad, col_shard.n_pad))
    return m.tocsc()[:, col_shard.lo:col_shard.hi].tocsr()


def halo_graphs(ui_local, iu_cols_local):
    """The item-side pair (A_ui[U_r, :] [per_u, I_pad], A_iu[:, U_r] [I_pad, per_u]) on COMPACT item ids: returns
    (need, A_ui[U_r, need] [per_u, n_need], A_iu[need, U_r] [n_need, per_u]) with need = the sorted global ids of the items
    this rank's edges reference (scheme "halo")."""
    ui_local = sp.csr_matrix(ui_local)
    need = np.unique(ui_local.indices).astype(np.int64)
    if need.shape[0] == 0:
        need = np.zeros(1, np.int64)
    return need, ui_local[:, need].tocsr(), sp.csr_matrix(iu_cols_local)[need, :].tocsr()


# ---------------------------------------------------------------------------------------------
# collectives with autograd
# ---------------------------------------------------------------------------------------------
def _solo(group):
    """world == 1: every collective is the identity; no RCCL launch (MMSSL_DIST_FORCE_COLLECTIVES=1 keeps the
    launches, e.g. to exercise RCCL-in-hipGraph capture on one GPU)."""
    return dist.get_world_size(group) == 1 and os.environ.get("MMSSL_DIST_FORCE_COLLECTIVES", "0") != "1"


# ---- peer exchange (mmssl_amd/peer.py, csrc/peer.hip): the same exchanges without a collective library in the data path ----
_PEER = {}


def _gkey(group):
    return group if group is not None else dist.group.WORLD


def enable_peer_exchange(group=None, device=None, timeout_ms=20000):
    """Route every exchange of `group` (all-gather / reduce-scatter of table rows, the small all-reduces) through
    IPC-mapped peer windows written and read by kernels. A host-side collective (IPC handles travel over `group`); the
    ranks must be processes of ONE node. Returns the PeerComm."""
    from .peer import PeerComm
    key = _gkey(group)
    if key not in _PEER:
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _PEER[key] = PeerComm(key, dev, timeout_ms=timeout_ms)
    return _PEER[key]


def disable_peer_exchange(group=None):
    pc = _PEER.pop(_gkey(group), None)
    if pc is not None:
        pc.close()


def _peer(group):
    return _PEER.get(_gkey(group)) if _PEER else None


def _all_reduce(t, group):
    pc = _peer(group)
    if pc is not None and t.is_cuda and not _solo(group):
        return pc.all_reduce_(t)
    if not _solo(group):
        fence = t.is_cuda and dist.get_backend(group) == "gloo"        # (test transport: see _gloo_device_fence)
        if fence:
            torch.cuda.synchronize(t.device)
        COMM["dist_calls"] += 1
        dist.all_reduce(t, group=group)
        if fence:
            torch.cuda.synchronize(t.device)
    return t


def _gloo_device_fence(t):
    """TEST TRANSPORT ONLY (ranks sharing one GPU talk through gloo, which stages device tensors through the host on its
    own streams): a device-wide synchronisation on both sides of the collective. With >= 4 ranks on one GPU and the item-side
    node's column-chunk lanes (collectives issued from 4 - 8 forked streams) torch's gloo path hands stale or half-written
    buffers to the consumers - repeated steps on unchanged inputs disagree, garbage in gradients, once a loss of 6e5 - while
    the same lanes are bit-stable over RCCL's stream-ordered collectives (world 1, forced launches, 1 / 2 / 4 chunks) and over
    gloo with this fence (world 4 / 8): profiles/NOTEBOOK.md round 5, tools/repeat_probe_synth.py. RCCL needs no fence."""
    if t.is_cuda:
        torch.cuda.synchronize(t.device)


def _all_gather_into(out, x, group):
    """all_gather_into_tensor; device tensors on a gloo group (tests: several ranks sharing ONE GPU) go through gloo's device
    all-reduce of a zero-padded buffer - gloo has no device all-gather - which is the same data movement seen from the
    step."""
    pc = _peer(group)
    if pc is not None and x.is_cuda:
        out.copy_(pc.gather(x if x.dim() == 2 else x.reshape(x.shape[0], -1)).view_as(out))
        return out
    if x.is_cuda and dist.get_backend(group) == "gloo":
        per = x.shape[0]
        r = dist.get_rank(group)
        out.zero_()
        out[r * per:(r + 1) * per].copy_(x)
        _gloo_device_fence(out)
        COMM["dist_calls"] += 1
        dist.all_reduce(out, group=group)
        _gloo_device_fence(out)
        return out
    COMM["dist_calls"] += 1
    dist.all_gather_into_tensor(out, x.contiguous(), group=group)
    return out


def _reduce_scatter_sum(full, per, group):
    if _solo(group):
        return full
    pc = _peer(group)
    if pc is not None and full.is_cuda:
        f2 = full if full.dim() == 2 else full.reshape(full.shape[0], -1)
        return pc.reduce(f2, per).view((per,) + tuple(full.shape[1:]))
    if dist.get_backend(group) == "gloo":      # gloo has no reduce_scatter: all-reduce + slice
        full = full.contiguous()
        _gloo_device_fence(full)
        COMM["dist_calls"] += 1
        dist.all_reduce(full, group=group)
        _gloo_device_fence(full)
        r = dist.get_rank(group)
        return full[r * per:(r + 1) * per].clone()
    out = torch.empty((per,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    COMM["dist_calls"] += 1
    dist.reduce_scatter_tensor(out, full.contiguous(), group=group)
    return out


class AllGatherRows(torch.autograd.Function):
    """[per, d] row shard -> [world*per, d]; backward = reduce-scatter(sum) of the full gradient."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group, ctx.per = group, x.shape[0]
        world = dist.get_world_size(group)
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        _all_gather_into(out, x.contiguous(), group)
        _log_comm("all_gather", out)
        return out

    @staticmethod
    def backward(ctx, g):
        _log_comm("reduce_scatter", g)
        return _reduce_scatter_sum(g, ctx.per, ctx.group), None


class ReduceScatterRows(torch.autograd.Function):
    """[world*per, d] partial products -> this rank's [per, d] rows of their sum; backward = all-gather of the gradient
    (the composed form of the item-side scheme: i_r = reduce_scatter(A_iu[:, U_r] . u_r))."""

    @staticmethod
    def forward(ctx, full, per, group):
        ctx.group = group
        _log_comm("reduce_scatter", full)
        return _reduce_scatter_sum(full.contiguous(), per, group)

    @staticmethod
    def backward(ctx, g):
        return _all_gather_raw(g.contiguous(), ctx.group), None, None


class GatherBatchRows(torch.autograd.Function):
    """Rows table[idx] of a row-sharded table, replicated on every rank. idx is global; each
    rank contributes the rows it owns (others are exact zeros) and the buffers are summed."""

    @staticmethod
    def forward(ctx, table, idx, lo, group):
        per = table.shape[0]
        mine = ((idx >= lo) & (idx < lo + per))
        local = (idx - lo).clamp(0, per - 1)
        rows = table[local] * mine.unsqueeze(1).to(table.dtype)
        _all_reduce(rows, group) if not _solo(group) else dist.all_reduce(rows, group=group)
        ctx.save_for_backward(local, mine)
        ctx.per = per
        return rows

    @staticmethod
    def backward(ctx, g):
        local, mine = ctx.saved_tensors
        out = torch.zeros((ctx.per, g.shape[1]), dtype=g.dtype, device=g.device)
        out.index_add_(0, local, g * mine.unsqueeze(1).to(g.dtype))
        return out, None, None, None


class GatherBatchRowsMulti(torch.autograd.Function):
    """Several (table, idx) row gathers with ONE all-reduce: the [B_k, d] pieces are packed into one
    buffer (each row has exactly one non-zero contributor, so the sum is exact). With a backend that has the
    owned-row kernels (HipBackend) every piece is ONE launch forward and one backward; the generic form (CPU / gloo
    tests) composes torch index ops."""

    @staticmethod
    def forward(ctx, group, n, bk, *args):
        tables, idxs, los = args[:n], args[n:2 * n], args[2 * n:3 * n]
        fast = bk is not None and hasattr(bk, "gather_owned") and tables[0].is_cuda
        sizes = [int(i.shape[0]) for i in idxs]
        if fast:
            d = tables[0].shape[1]
            packed = torch.empty((sum(sizes), d), dtype=tables[0].dtype, device=tables[0].device)
            o = 0
            for t, idx, lo, sz in zip(tables, idxs, los, sizes):
                bk.gather_owned(t.detach().contiguous(), idx, lo, packed[o:o + sz])
                o += sz
            ctx.meta = [(idx, lo, t.shape[0]) for t, idx, lo in zip(tables, idxs, los)]
            ctx.tkeys = [(t.data_ptr(), tuple(t.shape)) for t in tables]      # repeated tables share one gradient
        else:
            pieces, meta = [], []
            for t, idx, lo in zip(tables, idxs, los):
                per = t.shape[0]
                mine = (idx >= lo) & (idx < lo + per)
                local = (idx - lo).clamp(0, per - 1)
                pieces.append(t[local] * mine.unsqueeze(1).to(t.dtype))
                meta.append((local, mine, per))
            packed = torch.cat(pieces, 0)
            ctx.meta = meta
        _all_reduce(packed, group)
        _log_comm("all_reduce", packed)
        ctx.fast, ctx.bk = fast, bk
        return tuple(packed.split(sizes, 0))

    @staticmethod
    def backward(ctx, *gs):
        outs = []
        if ctx.fast:
            # ONE zero fill for all table gradients; pieces gathered from the same table (pos / neg items) scatter-add
            # into the same gradient, which is returned once (the other positions get None = zero)
            first, rows = {}, 0
            for k, (g, m) in enumerate(zip(gs, ctx.meta)):
                if g is not None and ctx.tkeys[k] not in first:
                    first[ctx.tkeys[k]] = (k, rows)
                    rows += m[2]
            live = [g for g in gs if g is not None]
            if not live:
                return (None, None, None) + (None,) * (3 * len(ctx.meta))
            # (zero-filled HERE, right in front of the scatter-adds: filled ahead of time - during the forward, on an idle
            # stream - the lines have left the L2s by now and the fp32 atomics run against memory: 27 - 51 us per launch
            # instead of 4 - 5, measured)
            buf = torch.zeros((rows, live[0].shape[1]), dtype=live[0].dtype, device=live[0].device)
            for k, (g, m) in enumerate(zip(gs, ctx.meta)):
                if g is None:
                    outs.append(None)
                    continue
                idx, lo, per = m
                k0, r0 = first[ctx.tkeys[k]]
                ctx.bk.scatter_owned(g.contiguous(), idx, lo, buf[r0:r0 + per])
                outs.append(buf[r0:r0 + per] if k == k0 else None)
            n = len(ctx.meta)
            return (None, None, None) + tuple(outs) + (None,) * (2 * n)
        for g, m in zip(gs, ctx.meta):
            if g is None:
                outs.append(None)
                continue
            if ctx.fast:
                idx, lo, per = m
                o = torch.zeros((per, g.shape[1]), dtype=g.dtype, device=g.device)
                ctx.bk.scatter_owned(g.contiguous(), idx, lo, o)
            else:
                local, mine, per = m
                o = torch.zeros((per, g.shape[1]), dtype=g.dtype, device=g.device)
                o.index_add_(0, local, g * mine.unsqueeze(1).to(g.dtype))
            outs.append(o)
        n = len(ctx.meta)
        return (None, None, None) + tuple(outs) + (None,) * (2 * n)


# ---------------------------------------------------------------------------------------------
# sharded model + step
# ---------------------------------------------------------------------------------------------
def _pad_rows(t, n_pad):
    """All rows of a global [n, ...] tensor, zero-padded to n_pad rows."""
    if t.shape[0] == n_pad:
        return t.clone()
    out = torch.zeros((n_pad,) + tuple(t.shape[1:]), dtype=t.dtype)
    out[:t.shape[0]] = t
    return out


def choose_replicate_feats(n_items_pad, feat_dims, d, world, link_gbs=60.0, tflops=90.0):
    """Should the item-side scheme replicate the feature matrices (ShardedMMSSL(replicate_feats=True))? Compares, per step
    and rank, the link time of the two collectives it removes - (N-1)/N of [I, nm d] fp32 each, spread over N-1 links -
    with the time of the extra projection flops (forward + weight gradient over all items instead of 1/N of them).
    configs[4] (two 128-wide features, d = 128, N = 8): 4.3 ms of link time against 1.3 ms of flops -> True; the Baby
    shape x 8 (4096 + 1024 wide): 0.3 ms against 1.9 ms -> False."""
    if world <= 1:
        return False
    nm = len(feat_dims)
    saved_s = 2.0 * n_items_pad * nm * d * 4.0 / world / (link_gbs * 1e9)           # per link: (N-1)/N * bytes / (N-1)
    extra_s = 2.0 * 2.0 * (world - 1.0) / world * n_items_pad * float(sum(feat_dims)) * d / (tflops * 1e12)
    return saved_s > 1.5 * extra_s


class ShardedMMSSL(nn.Module):
    """MMSSL.forward (reference Models.py:171-220) over row shards. Parameters:
    replicated  image_trans.{weight,bias}, text_trans.{weight,bias}, w_self_attention_cat
    sharded     user_id_embedding [per_u, d], item_id_embedding [per_i, d]."""

    def __init__(self, backend, cfg, ush, ish, state, image_feats, text_feats, group=None, scheme="gather-both", chunks=0,
                 replicate_feats=False):
        """scheme: which local graphs forward() receives as (ui, iu) and how the propagation communicates:
             "gather-both"  ui = A_ui[U_r, :], iu = A_iu[I_r, :] (shard_graph): all-gather of the item table before
                            A_ui . X_i and of the user table before A_iu . X_u
             "item-side"    ui = A_ui[U_r, :], iu = A_iu[:, U_r] (shard_graph_cols): only this rank's users' edges; every
                            collective is of item-table size (_ShardedItemSide)
             "halo"         the item-side graphs with COMPACT item columns (halo_graphs) + `model.halo` = their HaloPlan:
                            only the item rows this rank's edges reference travel (all-to-all of row lists)
           chunks: column chunks per collective of the item-side node (0 = by size, see n_chunks).
           replicate_feats (item-side scheme): every rank keeps the WHOLE constant feature matrices (Models.py:46-47) and
                            projects all items itself, so the projected features never travel: the modal chain's all-gather
                            of X [I, nm d] and its adjoint reduce-scatter disappear (its other two collectives stay) - 20 % of
                            a step's bytes for configs[4] - for N times the projection flops per rank (the weight gradient
                            each rank then forms from its users' partial g(X) is summed by the all-reduce the replicated
                            parameters' gradients take anyway). Worth it when the features are narrow
                            (choose_replicate_feats); injected / drawn dropout masks then cover ALL item rows and must be
                            the same on every rank."""
        super().__init__()
        self.bk, self.cfg, self.ush, self.ish, self.group = backend, cfg, ush, ish, group
        self.scheme, self.chunks = scheme, int(chunks)
        self.replicate_feats = bool(replicate_feats)
        if self.replicate_feats and scheme != "item-side":
            raise ValueError("replicate_feats needs scheme='item-side'")
        self.img_w = nn.Parameter(state["image_trans.weight"].clone())
        self.img_b = nn.Parameter(state["image_trans.bias"].clone())
        self.txt_w = nn.Parameter(state["text_trans.weight"].clone())
        self.txt_b = nn.Parameter(state["text_trans.bias"].clone())
        self.w_cat = nn.Parameter(state["weight_dict.w_self_attention_cat"].clone())
        self.E_u = nn.Parameter(ush.slice_rows(state["user_id_embedding.weight"]))
        self.E_i = nn.Parameter(ish.slice_rows(state["item_id_embedding.weight"]))
        rows = (lambda t: _pad_rows(torch.as_tensor(t), ish.n_pad)) if self.replicate_feats else ish.slice_rows
        self.register_buffer("image_feats", rows(torch.as_tensor(image_feats)), persistent=False)
        self.register_buffer("text_feats", rows(torch.as_tensor(text_feats)), persistent=False)

    def _forward_fused(self, graphs, keep_masks, modal_empty):
        bk, c = self.bk, self.cfg
        ui, iu, img_ui, img_iu, txt_ui, txt_iu = graphs
        scale, keep = 1.0, None
        if self.training and c.drop_rate > 0:
            scale = 1.0 / (1.0 - c.drop_rate)
            if keep_masks is not None:           # injected (parity runs): [2, per_items, d] as the projection takes it
                keep = torch.stack(tuple(keep_masks)) if isinstance(keep_masks, (tuple, list)) else keep_masks
            else:                                # drawn by the projection's epilogue from the device generator
                keep = ("draw", float(c.drop_rate), bool(getattr(self, "_external_ticks", False)))
                if (getattr(self, "replicate_feats", False) and not _solo(self.group) and hasattr(bk, "ops")
                        and not getattr(self, "_rng_checked", False)):
                    # replicated features: every rank projects ALL items and must drop the SAME entries, i.e. draw from
                    # the same generator state (seed and launch counter). Checked once, on the host, before the first draw:
                    # ranks seeded per rank (2022 + rank, right for the sharded features) would project different X and
                    # all-reduce inconsistent weight gradients without any error
                    mine = [int(v) for v in bk.ops._rng_state(self.E_i.device).cpu().tolist()]
                    got = [None] * dist.get_world_size(self.group)
                    dist.all_gather_object(got, mine, group=self.group)
                    if any(g_ != got[0] for g_ in got):
                        raise RuntimeError("ShardedMMSSL(replicate_feats=True): the ranks' dropout generators differ (%s): "
                                           "seed them identically (ops.seed_dropout(seed) with ONE seed on every rank) or "
                                           "inject keep_masks" % (got,))
                    self._rng_checked = True
        if modal_empty:
            z = getattr(self, "_zero_views", None)
            if z is None or z[0].device != self.E_u.device:
                z = self._zero_views = (torch.zeros_like(self.E_u), torch.zeros_like(self.E_i))
            img_uid = txt_uid = z[0]
            img_iid = txt_iid = z[1]
            if hasattr(bk, "ops"):       # zero (not missing) gradient for w_cat, without arithmetic on the tables
                u, i = bk.ops.zero_grad_anchor(self.E_u, self.w_cat), self.E_i
            else:
                u, i = self.E_u + 0 * self.w_cat.sum(), self.E_i
        else:
            Ei_full, Eu_full = self._gather(self.E_i), self._gather(self.E_u)
            img_uid, img_iid = bk.spmm(img_ui, Ei_full), bk.spmm(img_iu, Eu_full)
            txt_uid, txt_iid = bk.spmm(txt_ui, Ei_full), bk.spmm(txt_iu, Eu_full)
            u = bk.l2norm_rows(self._fusion(img_uid, txt_uid), self.E_u, c.id_cat_rate)
            i = bk.l2norm_rows(self._fusion(img_iid, txt_iid), self.E_i, c.id_cat_rate)
        d = c.embed_size
        if getattr(self, "scheme", "gather-both") in ("item-side", "halo") and not _solo(self.group):
            xch = getattr(self, "halo", None) if self.scheme == "halo" else None
            if self.scheme == "halo" and xch is None:
                raise RuntimeError("ShardedMMSSL(scheme='halo') needs model.halo = HaloPlan(...) (see halo_graphs)")
            u_g, i_g, ss, MI, MU = _ShardedItemSide.apply(
                2, scale, keep, ui, iu, c.n_ui_layers, c.model_cat_rate, bk, self.group, self.n_chunks(2), xch,
                bool(getattr(self, "replicate_feats", False)), u, i,
                self.image_feats, self.text_feats, self.img_w, self.txt_w, self.img_b, self.txt_b)
        else:       # (one rank without forced collectives: both schemes are the same computation)
            if getattr(self, "replicate_feats", False) and self.image_feats.shape[0] != self.E_i.shape[0]:
                raise RuntimeError("replicate_feats on one rank: the features must have the rank's (= all) item rows")
            if getattr(self, "scheme", "") == "halo" and ui.shape[1] != self.ish.n_pad:
                raise RuntimeError("scheme 'halo' on one rank without collectives: pass the item-side graphs (full item columns)")
            u_g, i_g, ss, MI, MU = _ShardedHotForward.apply(
                2, scale, keep, ui, iu, c.n_ui_layers, c.model_cat_rate, bk, self.group, u, i,
                self.image_feats, self.text_feats, self.img_w, self.txt_w, self.img_b, self.txt_b)
        self._feat_ss_local = ss
        img_item, txt_item, img_user, txt_user = MI[:, :d], MI[:, d:], MU[:, :d], MU[:, d:]
        return (u_g, i_g, img_item, txt_item, img_user, txt_user, u_g, i_g, img_uid, txt_uid, img_iid, txt_iid)

    def replicated_parameters(self):
        return [self.img_w, self.img_b, self.txt_w, self.txt_b, self.w_cat]

    CHUNK_BYTES = 64 << 20

    def n_chunks(self, nm):
        """Column chunks per collective of the item-side node. Explicit (`chunks` > 0) or by size: collectives below
        CHUNK_BYTES (the whole gathered item table at width d) are latency-bound and stay whole - every chunk is one more
        RCCL launch - larger ones are cut in two, from 4 x CHUNK_BYTES on (configs[4]: a 512 MB table per collective) in four:
        only the first chunk's transfer and the last chunk's product stay exposed per collective (tools/shard_bytes.py:
        4 chunks are what brings configs[4]'s link-bound step from 3.8x to 5.9x one GPU at 70 GB/s per link)."""
        d = self.cfg.embed_size
        ok = getattr(self.bk, "chunk_ok", lambda w, n: w % n == 0)
        nc = int(getattr(self, "chunks", 0))
        if nc <= 0:
            # (halo: the JOB's largest referenced-row count, identical on every rank - n_need itself is per rank, and two
            # ranks either side of a threshold would issue different numbers of exchanges of different widths)
            rows = self.halo.n_need_max if (getattr(self, "scheme", "") == "halo" and getattr(self, "halo", None)) else self.ish.n_pad
            full = rows * d * 4
            nc = 4 if full >= 4 * self.CHUNK_BYTES else (2 if full >= self.CHUNK_BYTES else 1)
        while nc > 1 and not (ok(d, nc) and ok(nm * d, nc)):
            nc -= 1
        return max(nc, 1)

    def _gather(self, x):
        return AllGatherRows.apply(x, self.group)

    def _fusion(self, a, b):
        c = self.cfg
        fold = self.w_cat.view(c.head_num, c.embed_size, c.embed_size).sum(0)
        return self.bk.linear(a + b, (0.5 * fold).t().contiguous())

    def forward(self, graphs, keep_masks=None, modal_empty=False, fused=True):
        """graphs = local (ui, iu, img_ui, img_iu, txt_ui, txt_iu). Returns the local rows of the
        reference's 12 outputs (0/6 and 1/7 identical). `fused` selects the single-node
        implementation (_ShardedHotForward); fused=False composes differentiable backend ops and
        AllGatherRows (the same math, kept as the cross-check)."""
        bk, c = self.bk, self.cfg
        pc = _peer(self.group)
        if pc is not None and self.E_i.is_cuda and not _solo(self.group):
            pc.begin_step()          # every rank has finished the previous step: the exchange windows may be rewritten
        self.last_fused = bool(fused and bk.packed_supported([self.image_feats.shape[1], self.text_feats.shape[1]],
                                                             self.ish.per, c.embed_size))
        if self.last_fused:
            return self._forward_fused(graphs, keep_masks, modal_empty)
        # (feature widths the grouped projection does not take: the composed form below runs them)
        if getattr(self, "replicate_feats", False):
            raise RuntimeError("replicate_feats runs on the packed node only (feature widths %s, d = %d are not packable)" % (
                [self.image_feats.shape[1], self.text_feats.shape[1]], c.embed_size))
        if getattr(self, "scheme", "gather-both") == "halo":
            raise RuntimeError("ShardedMMSSL(scheme='halo') runs on the packed node only (feature widths %s, d = %d are not "
                               "packable): use scheme='item-side' for this model" % (
                                   [self.image_feats.shape[1], self.text_feats.shape[1]], c.embed_size))
        ui, iu, img_ui, img_iu, txt_ui, txt_iu = graphs
        scale = 1.0
        km_i = km_t = None
        if self.training and c.drop_rate > 0:
            scale = 1.0 / (1.0 - c.drop_rate)
            if keep_masks is not None:
                km_i, km_t = keep_masks
            else:
                shape = (self.ish.per, c.embed_size)
                km_i = (torch.rand(shape, device=self.E_i.device) >= c.drop_rate).to(torch.uint8)
                km_t = (torch.rand(shape, device=self.E_i.device) >= c.drop_rate).to(torch.uint8)
        x_img = bk.linear(self.image_feats, self.img_w, self.img_b, km_i, scale)
        x_txt = bk.linear(self.text_feats, self.txt_w, self.txt_b, km_t, scale)
        # one gather for both modalities (columns concatenated): halves the collective count
        d = c.embed_size
        item_side = getattr(self, "scheme", "gather-both") == "item-side"
        if item_side:        # A_iu . X_u = reduce_scatter(A_iu[:, U_r] . X_u_r): `iu` is the column block of A_iu
            to_items = lambda x, epi=bk.EPI_NONE: ReduceScatterRows.apply(bk.spmm(iu, x), self.ish.per, self.group)   # noqa: E731
        else:
            to_items = lambda x, epi=bk.EPI_NONE: bk.spmm(iu, self._gather(x), epi)                                   # noqa: E731
        x_both = self._gather(torch.cat((x_img, x_txt), 1))
        img_user = bk.spmm(ui, x_both[:, :d].contiguous())
        txt_user = bk.spmm(ui, x_both[:, d:].contiguous())
        if item_side:
            both = to_items(torch.cat((img_user, txt_user), 1))
            img_item, txt_item = both[:, :d], both[:, d:]
        else:
            u_both = self._gather(torch.cat((img_user, txt_user), 1))
            img_item = bk.spmm(iu, u_both[:, :d].contiguous())
            txt_item = bk.spmm(iu, u_both[:, d:].contiguous())
        if modal_empty:
            # empty modal graphs (the reference's steady state): the id views are exact zeros
            zu = torch.zeros_like(self.E_u)
            zi = torch.zeros_like(self.E_i)
            img_uid = txt_uid = zu
            img_iid = txt_iid = zi
            u = self.E_u + 0 * zu
            i = self.E_i + 0 * zi
        else:
            Ei_full, Eu_full = self._gather(self.E_i), self._gather(self.E_u)
            img_uid, img_iid = bk.spmm(img_ui, Ei_full), bk.spmm(img_iu, Eu_full)
            txt_uid, txt_iid = bk.spmm(txt_ui, Ei_full), bk.spmm(txt_iu, Eu_full)
            u = bk.l2norm_rows(self._fusion(img_uid, txt_uid), self.E_u, c.id_cat_rate)
            i = bk.l2norm_rows(self._fusion(img_iid, txt_iid), self.E_i, c.id_cat_rate)
        u_sum, i_sum = u, i
        for layer in range(c.n_ui_layers):
            epi = bk.EPI_SOFTMAX if layer == c.n_ui_layers - 1 else bk.EPI_NONE
            u = bk.spmm(ui, self._gather(i), epi)
            i = to_items(u, epi)
            if item_side and epi == bk.EPI_SOFTMAX:
                i = bk.softmax_rows(i)           # whole rows exist only after the reduce-scatter
            u_sum = u_sum + u
            i_sum = i_sum + i
        inv = 1.0 / (c.n_ui_layers + 1)
        r = c.model_cat_rate
        u_g = bk.l2norm_rows(txt_user, bk.l2norm_rows(img_user, u_sum * inv, r), r)
        i_g = bk.l2norm_rows(txt_item, bk.l2norm_rows(img_item, i_sum * inv, r), r)
        return (u_g, i_g, img_item, txt_item, img_user, txt_user, u_g, i_g, img_uid, txt_uid, img_iid, txt_iid)



# launch accounting for bench.py: every collective of a counted step {kind, bytes of the full (gathered /
# to-be-scattered / reduced) buffer}; COMM["log"] is None when not recording
# log: per-step accounting of the exchanges; dist_calls: torch.distributed collectives that carried a tensor in the data
# path (0 per step when the peer exchange is the transport)
COMM = {"log": None, "dist_calls": 0}


def _log_comm(kind, t):
    if COMM["log"] is not None:
        COMM["log"].append((kind, tuple(t.shape), t.numel() * t.element_size()))


def _all_gather_raw(x, group):
    world = dist.get_world_size(group)
    if _solo(group):
        _log_comm("all_gather", x)
        return x
    pc = _peer(group)
    if pc is not None and x.is_cuda and x.dim() == 2 and x.stride(1) == 1:
        out = pc.gather(x)                    # the window itself: no staging copy, row-pitched shards pushed as they are
        _log_comm("peer_gather", out)
        return out
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    _all_gather_into(out, x.contiguous(), group)
    _log_comm("all_gather", out)
    return out


def _coalesced(group, device):
    """One grouped launch for the collectives issued inside (RCCL: ncclGroupStart / End); None where the backend has no
    grouped form (gloo: the calls simply run one after the other)."""
    import contextlib
    if dist.get_backend(group) != "nccl" or not hasattr(dist, "_coalescing_manager"):
        return contextlib.nullcontext(), False
    return dist._coalescing_manager(group=group, device=device, async_ops=False), True


def _all_gather_pair(xa, xb, group):
    """Row shards xa, xb (same sharding, different widths) -> their gathered forms, as ONE grouped launch."""
    if _solo(group):
        if COMM["log"] is not None:
            COMM["log"].append(("all_gather", (tuple(xa.shape), tuple(xb.shape)), 4 * (xa.numel() + xb.numel())))
        return xa, xb
    if _peer(group) is not None and xa.is_cuda:
        return _all_gather_raw(xa, group), _all_gather_raw(xb, group)
    world = dist.get_world_size(group)
    outs = [torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device) for x in (xa, xb)]
    cm, grouped = _coalesced(group, xa.device)
    with cm:
        for o, x in zip(outs, (xa, xb)):
            _all_gather_into(o, x.contiguous(), group)
    if COMM["log"] is not None:
        if grouped:
            COMM["log"].append(("all_gather", tuple(tuple(o.shape) for o in outs), 4 * sum(o.numel() for o in outs)))
        else:
            for o in outs:
                _log_comm("all_gather", o)
    return outs[0], outs[1]


def _reduce_scatter_pair(fa, fb, per, group):
    """Full partial products fa, fb (same row count) -> this rank's rows of their sums, as ONE grouped launch."""
    if _solo(group) or dist.get_backend(group) == "gloo" or _peer(group) is not None:
        outs = (_reduce_scatter_sum(fa, per, group), _reduce_scatter_sum(fb, per, group))
        if COMM["log"] is not None:
            if _peer(group) is not None and not _solo(group):
                _log_comm("peer_reduce", fa)
                _log_comm("peer_reduce", fb)
            elif _solo(group):
                COMM["log"].append(("reduce_scatter", (tuple(fa.shape), tuple(fb.shape)), 4 * (fa.numel() + fb.numel())))
            else:
                _log_comm("reduce_scatter", fa)
                _log_comm("reduce_scatter", fb)
        return outs
    outs = [torch.empty((per,) + tuple(f.shape[1:]), dtype=f.dtype, device=f.device) for f in (fa, fb)]
    cm, grouped = _coalesced(group, fa.device)
    with cm:
        for o, f in zip(outs, (fa, fb)):
            COMM["dist_calls"] += 1
            dist.reduce_scatter_tensor(o, f.contiguous(), group=group)
    if COMM["log"] is not None:
        if grouped:
            COMM["log"].append(("reduce_scatter", (tuple(fa.shape), tuple(fb.shape)), 4 * (fa.numel() + fb.numel())))
        else:
            _log_comm("reduce_scatter", fa)
            _log_comm("reduce_scatter", fb)
    return outs[0], outs[1]


class _Side:
    """The GCN chain's side stream next to the current one (None on CPU: everything runs inline)."""

    def __init__(self, bk, ref):
        side = bk.side_streams(ref.device) if (hasattr(bk, "side_streams") and ref.is_cuda) else None
        self.side = side[2] if side else None
        self.main = torch.cuda.current_stream(ref.device) if side else None

    def gcn(self):
        import contextlib
        return torch.cuda.stream(self.side) if self.side is not None else contextlib.nullcontext()

    def fork(self):
        if self.side is not None:
            self.side.wait_stream(self.main)

    def join(self):
        if self.side is not None:
            self.main.wait_stream(self.side)

    def meet(self, solo):
        """Both chains' operands of one GROUPED collective must exist on the stream that issues it (the current one);
        afterwards the side stream continues behind it. One rank without forced collectives: nothing is issued, the
        chains stay independent."""
        if self.side is not None and not solo:
            self.main.wait_stream(self.side)

    def part(self, solo):
        if self.side is not None and not solo:
            self.side.wait_stream(self.main)

    def lend(self, t, solo):
        """A grouped collective's output (allocated on the current stream) that the SIDE stream consumes: tell the
        allocator, or the block could be handed out again on the current stream while the side stream still reads it."""
        if self.side is not None and not solo and t.is_cuda:
            t.record_stream(self.side)


class _ShardedHotForward(torch.autograd.Function):
    """Row-sharded counterpart of hotnode._HotNode, one autograd node: the grouped projection of the local item rows,
    the PACKED modal chain (all modalities side by side: one d = 64 nm SpMM pair instead of nm of them), the GCN chain
    on a side stream and the two-sided fuse kernels - with an all-gather of the row shards before every A_r . X and, in
    the hand-written backward, a reduce-scatter of every A_r^T . gY_r.

    Collectives per step (L GCN layers): forward 2L gathers for the GCN chain; the modal chain's two gathers are GROUPED
    with the GCN gathers of the same sharding (items / users) at the point where the projection has typically finished
    (GCN gathers 2 and 3 for L >= 2: the first GCN layer runs under the GEMM), so they cost no launch of their own.
    Backward: the modal chain's two reduce-scatters are on the critical path (the weight gradient waits for them) and are
    grouped with the FIRST two GCN reduce-scatters. 2L + 2L collectives instead of (2L + 2 nm) twice."""

    @staticmethod
    def forward(ctx, nm, scale, keep, ui, iu, n_layers, r, bk, group, u0, i0, *flat):
        Fs, Ws, bs = flat[:nm], flat[nm:2 * nm], flat[2 * nm:3 * nm]
        draw_p, ext_tick = 0.0, False
        if isinstance(keep, tuple):                  # ("draw", p, external_tick): fresh masks from the device generator
            _, draw_p, ext_tick = keep
            keep = None
        g = group
        solo = _solo(g)
        st = _Side(bk, u0)
        twin = (lambda p, k: p.twin(k)) if hasattr(ui, "twin") else (lambda p, k: p)
        st.fork()
        X, keep = bk.proj_forward(list(Fs), list(Ws), list(bs), keep, scale, draw_p, ext_tick)      # [per_i, 64 nm]
        pair_at = min(2, 2 * n_layers - 2)           # GCN gather (even = item rows) the modal chain's first one rides on
        us, its = [u0], [i0]
        u, i = u0, i0
        MU = MI = None
        k = 0
        for l in range(n_layers):
            epi = bk.EPI_SOFTMAX if l == n_layers - 1 else bk.EPI_NONE
            # ---- item rows -> user rows
            if k == pair_at:
                st.meet(solo)
                i_full, X_full = _all_gather_pair(i, X, g)
                st.part(solo)
                st.lend(i_full, solo)
                MU = bk.spmm_raw(ui, False, X_full, bk.EPI_NONE)
            else:
                with st.gcn():
                    i_full = _all_gather_raw(i, g)
            with st.gcn():
                u = bk.spmm_raw(twin(ui, 2), False, i_full, epi)
            k += 1
            # ---- user rows -> item rows
            if k == pair_at + 1:
                st.meet(solo)
                u_full, MU_full = _all_gather_pair(u, MU, g)
                st.part(solo)
                st.lend(u_full, solo)
                MI = bk.spmm_raw(iu, False, MU_full, bk.EPI_NONE)
            else:
                with st.gcn():
                    u_full = _all_gather_raw(u, g)
            with st.gcn():
                i = bk.spmm_raw(twin(iu, 2), False, u_full, epi)
            k += 1
            us.append(u)
            its.append(i)
        st.join()
        inv = 1.0 / (n_layers + 1)
        u_g, i_g, ss = bk.fuse_fwd(us, MU, its, MI, inv, nm, r)
        ctx.save_for_backward(MU, MI, us[-1], its[-1], keep, *Fs)
        ctx.cfg = (nm, float(scale), ui, iu, n_layers, float(r), inv, bk, g, [b is not None for b in bs])
        ctx.set_materialize_grads(False)
        return u_g, i_g, ss, MI, MU

    @staticmethod
    def backward(ctx, Gu, Gi, g_ss, G_MI, G_MU):
        MU, MI, uG, iG, keep = ctx.saved_tensors[:5]
        Fs = ctx.saved_tensors[5:]
        nm, scale, ui, iu, n_layers, r, inv, bk, g, has_b = ctx.cfg
        per_u, per_i = MU.shape[0], MI.shape[0]
        d = uG.shape[1]
        Gu = Gu.contiguous() if Gu is not None else torch.zeros_like(uG)
        Gi = Gi.contiguous() if Gi is not None else torch.zeros_like(iG)
        g_ss = g_ss.contiguous().to(torch.float32) if g_ss is not None else None
        G_MI = G_MI.contiguous() if G_MI is not None else None
        G_MU = G_MU.contiguous() if G_MU is not None else None
        solo = _solo(g)
        fused_epi = solo and hasattr(bk, "ops")      # one rank: the scatter is the identity, the add rides in the SpMM store
        st = _Side(bk, Gu)
        twin = (lambda p, k: p.twin(k)) if hasattr(ui, "twin") else (lambda p, k: p)
        st.fork()
        gMU, g_u0, gMI = bk.fuse_bwd(MU, Gu, G_MU, MI, Gi, G_MI, nm, r, inv, g_ss)
        if st.side is not None:
            # from here on the gradient of u_0 exists; a step object may update the (sharded, never all-reduced) embedding
            # tables on the side stream as soon as the GCN chain there has produced the gradient of i_0
            bk.after_fuse_bwd = st.main.record_event()
            bk.tables_stream = st.side
            g_u0.record_stream(st.side)
            for t in (uG, iG, Gu, Gi):
                t.record_stream(st.side)

        def rs_add(plan, x, per, Z, alpha, epi=None, S=None, log=True):
            """reduce_scatter(A_r^T . x) + alpha * Z (then the softmax backward against S when epi asks for it)."""
            if fused_epi:
                if log:
                    _log_comm("reduce_scatter", Z)
                if epi is not None:
                    return bk.spmm_raw(plan, True, x, epi, Z, alpha, S)
                return bk.spmm_raw(plan, True, x, bk.EPI_AXPY, Z, alpha)
            part = bk.spmm_raw(plan, True, x, bk.EPI_NONE)
            _log_comm("reduce_scatter", part)
            y = _reduce_scatter_sum(part, per, g).add_(Z, alpha=alpha)
            return bk.softmax_rows_bwd(S, y, 1.0) if S is not None else y

        # ---- depth 1 (user rows): GCN A_iu_r^T g(i_G) and modal A_iu_r^T g(MI), one grouped reduce-scatter
        with st.gcn():
            gi = bk.softmax_rows_bwd(iG, Gi, inv)
        if fused_epi:
            with st.gcn():
                gu = rs_add(twin(iu, 2), gi, per_u, Gu, inv, bk.EPI_AXPY_SOFTMAX_BWD, uG, log=False)
            t = bk.spmm_raw(iu, True, gMI, bk.EPI_AXPY, gMU, 1.0)
            if COMM["log"] is not None:
                COMM["log"].append(("reduce_scatter", (tuple(Gu.shape), tuple(gMU.shape)), 4 * (Gu.numel() + gMU.numel())))
        else:
            with st.gcn():
                part_g = bk.spmm_raw(twin(iu, 2), True, gi, bk.EPI_NONE)
            part_m = bk.spmm_raw(iu, True, gMI, bk.EPI_NONE)
            st.meet(solo)
            rg, rm = _reduce_scatter_pair(part_g, part_m, per_u, g)
            st.part(solo)
            st.lend(rg, solo)
            t = rm.add_(gMU)
            with st.gcn():
                gu = bk.softmax_rows_bwd(uG, rg.add_(Gu, alpha=inv), 1.0)
        # ---- depth 2 (item rows): GCN A_ui_r^T g(u) and modal A_ui_r^T t (+ the projection's dropout backward)
        if fused_epi:
            with st.gcn():
                gi = rs_add(twin(ui, 2), gu, per_i, Gi, inv, log=False)
            if COMM["log"] is not None:
                COMM["log"].append(("reduce_scatter", (tuple(Gi.shape), tuple(gMI.shape)), 4 * (Gi.numel() + gMI.numel())))
            gX = bk.spmm_mask(ui, t, keep, d, scale) if keep is not None else bk.spmm_raw(ui, True, t, bk.EPI_NONE)
        else:
            with st.gcn():
                part_g = bk.spmm_raw(twin(ui, 2), True, gu, bk.EPI_NONE)
            part_m = bk.spmm_raw(ui, True, t, bk.EPI_NONE)
            st.meet(solo)
            rg, gX = _reduce_scatter_pair(part_g, part_m, per_i, g)
            st.part(solo)
            st.lend(rg, solo)
            if keep is not None:
                gX = bk.mask_packed(gX, keep, d, scale)
            with st.gcn():
                gi = rg.add_(Gi, alpha=inv)
        # ---- the weight gradient (current stream) next to the rest of the GCN chain (side stream)
        # (Round 5 joined the chains here for the per-modality kernels: one run in three or four at configs[4]'s size had a
        # text-projection weight gradient 6e-3 off. Root cause, round 6: csrc/linear.hip's wgrad10_kernel copied a register
        # whose load was still in flight - see the note in that kernel and tools/vmcnt_check.py. The overlap is back.)
        gW, gb = bk.proj_wgrad(gX, list(Fs), any(has_b))
        with st.gcn():
            for _ in range(n_layers - 1):
                gu = rs_add(twin(iu, 2), gi, per_u, Gu, inv)
                gi = rs_add(twin(ui, 2), gu, per_i, Gi, inv)
        st.join()
        if st.side is not None:
            gi.record_stream(st.main)
        grads_b = [(gb[k] if (gb is not None and has_b[k]) else None) for k in range(nm)]
        bk.table_grads = (g_u0.data_ptr(), gi.data_ptr())
        return (None,) * 9 + (g_u0, gi) + (None,) * nm + tuple(gW) + tuple(grads_b)


def _reduce_scatter_raw(full, per, group):
    _log_comm("peer_reduce" if (_peer(group) is not None and not _solo(group)) else "reduce_scatter", full)
    return _reduce_scatter_sum(full, per, group)


class _Lanes:
    """Column-chunk lanes. A d-wide propagation is nc independent d/nc-wide ones (LightGCN layers have no cross-column
    term, Models.py:201-211): lane c's whole chain - collective, product, product, collective, ... - is issued on stream c.
    RCCL runs the collectives of one process group on its own stream in ISSUE order, so while lane c's product runs, lane
    c+1's collective is on the links. Lanes meet only where a row is needed whole (the last layer's softmax) and at the
    end. On CPU (gloo tests) and for nc == 1 the lanes are plain loops."""

    def __init__(self, bk, ref, n, base=0, first_is_current=False, one_stream=False, via_main=False):
        """one_stream: all n lanes share ONE forked stream (chunks run one after the other there). via_main: a lane's plain
        collectives are issued from the origin stream (see coll)."""
        self.n = int(n)
        self.cuda = bool(ref.is_cuda and hasattr(bk, "lane_streams"))
        self.main, self.streams = None, [None] * self.n
        self.via_main = bool(via_main)
        if self.cuda:
            self.main = torch.cuda.current_stream(ref.device)
            pool = bk.lane_streams(ref.device, base + self.n)
            self.streams = [self.main if (first_is_current and c == 0) else pool[base + (0 if one_stream else c)]
                            for c in range(self.n)]

    def on(self, c):
        import contextlib
        return torch.cuda.stream(self.streams[c]) if self.cuda else contextlib.nullcontext()

    def fork(self):
        if self.cuda:
            for st in self.streams:
                if st is not self.main:
                    st.wait_stream(self.main)

    def join(self):
        if self.cuda:
            for st in self.streams:
                if st is not self.main:
                    self.main.wait_stream(st)

    def coll(self, c, fn, reads=()):
        """One plain collective of lane c (`reads`: lane-allocated tensors it consumes): issued on the lane, or - via_main -
        from the origin stream with the lane waiting for ITS event (exchanges that cannot be captured on a forked stream)."""
        if self.cuda and self.via_main and self.streams[c] is not self.main:
            self.to_main(c)
            for t in reads:
                t.record_stream(self.main)
            out = fn()
            self.after(c, self.mark())
            self.uses(out, [c])
            return out
        with self.on(c):
            return fn()

    def to_main(self, c):
        """The origin stream waits for lane c (a grouped collective is issued from the origin stream)."""
        if self.cuda and self.streams[c] is not self.main:
            self.main.wait_stream(self.streams[c])

    def after(self, c, ev):
        if self.cuda and ev is not None and self.streams[c] is not self.main:
            self.streams[c].wait_event(ev)

    def mark(self):
        return self.main.record_event() if self.cuda else None

    def meet(self):
        """Lane 0 waits for all lanes (then runs the whole-row kernel); `part` lets the others continue behind it. Side
        lanes synchronise THROUGH the origin stream: a direct wait between two forked streams of a hipGraph capture
        crashes hipStreamEndCapture (ROCm 7.2; bisected on the GPU, tools/README.md)."""
        if not self.cuda or self.n == 1 or all(st is self.streams[0] for st in self.streams):
            return
        if self.streams[0] is self.main:
            for st in self.streams[1:]:
                self.main.wait_stream(st)
            return
        for st in self.streams:
            self.main.wait_stream(st)
        self.streams[0].wait_stream(self.main)

    def part(self):
        if not self.cuda or self.n == 1 or all(st is self.streams[0] for st in self.streams):
            return
        if self.streams[0] is not self.main:
            self.main.wait_stream(self.streams[0])
        for st in self.streams[1:]:
            st.wait_stream(self.main)

    def uses(self, t, lanes=None):
        """`t` (allocated on the current stream) is read / written on the lanes: tell the caching allocator."""
        if self.cuda and t is not None and t.is_cuda:
            for st in (self.streams if lanes is None else [self.streams[c] for c in lanes]):
                if st is not self.main:
                    t.record_stream(st)


def _chunks_of(t, nc):
    """Column chunks of a row-major [rows, w] tensor as views (row pitch w)."""
    w = t.shape[1] // nc
    return [t[:, c * w:(c + 1) * w] for c in range(nc)] if nc > 1 else [t]


def _contig_chunks(t, nc):
    return [x.contiguous() for x in _chunks_of(t, nc)]



class _TableExchange:
    """How the item-side node moves item-table rows between ranks: the WHOLE table by RCCL all-gather / reduce-scatter
    (scheme "item-side"). gather: this rank's rows [per_i, w] -> what the local products gather from ([I_pad, w]);
    reduce: local partial products over that row space -> this rank's rows of their sum over all ranks."""

    def __init__(self, group, per_i):
        self.group, self.per_i = group, per_i

    def gather(self, x):
        return _all_gather_raw(x, self.group)

    def partial(self, rows, width):
        """Where the next [rows, width] partial product should be written (the peer exchange's window: the reduce then
        pulls from it in place), or None = allocate as usual."""
        pc = _peer(self.group)
        return pc.partial(rows, width) if (pc is not None and not _solo(self.group)) else None

    def reduce(self, P):
        return _reduce_scatter_raw(P, self.per_i, self.group)

    def gather_pair(self, a, b):
        return _all_gather_pair(a, b, self.group)

    def reduce_pair(self, fa, fb):
        return _reduce_scatter_pair(fa, fb, self.per_i, self.group)


def _all_to_all_rows(out, inp, out_rows, in_rows, group):
    """Variable all-to-all of row blocks (rank q gets inp's block q, sends its blocks likewise). Device tensors on a gloo
    group (tests: ranks sharing one GPU) are staged through the host."""
    if _peer(group) is not None and inp.is_cuda:
        raise RuntimeError("the halo scheme's all-to-alls run over RCCL: use scheme 'item-side' with the peer exchange")
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        o = torch.empty(out.shape, dtype=out.dtype)
        COMM["dist_calls"] += 1
        dist.all_to_all_single(o, inp.cpu(), output_split_sizes=out_rows, input_split_sizes=in_rows, group=group)
        out.copy_(o)
        return out
    COMM["dist_calls"] += 1
    dist.all_to_all_single(out, inp.contiguous(), output_split_sizes=out_rows, input_split_sizes=in_rows, group=group)
    return out


class HaloPlan:
    """Scheme "halo": a rank exchanges only the item rows its own users' edges REFERENCE. Built once per graph on the host:
      need        sorted global ids of the items this rank's edges touch (n_need of them: 55 % of the table for the
                  Baby-shaped weak-scaling graph at N = 8, 98 % for configs[4], a few % for a community-partitioned graph)
      the local graphs use COMPACT columns 0 .. n_need - 1 (A_ui[U_r, need], A_iu[need, U_r])
      send_idx    for every peer q, the local rows of MY item block that q needs (owners send exactly those rows)
      sel         [per_i, n_send] 0/1 selection matrix as a graph plan: summing the partial-product rows that came back for
                  my items is one deterministic SpMM (a row's contributions in rank order)
    gather = pack (row gather kernel) + all-to-all;  reduce = all-to-all + selection SpMM; each the other's adjoint."""

    def __init__(self, need, ish, group, bk, device):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        need = np.ascontiguousarray(need, dtype=np.int64)
        lists = [None] * world
        dist.all_gather_object(lists, need, group=group)           # host, once per graph
        owner_cut = np.searchsorted(need, np.arange(world + 1) * ish.per)
        self.recv_rows = [int(owner_cut[q + 1] - owner_cut[q]) for q in range(world)]      # what I get from owner q
        send = []
        for q in range(world):                                       # rows of MY block that rank q needs, as local ids
            nq = lists[q]
            a, b = np.searchsorted(nq, ish.lo), np.searchsorted(nq, ish.lo + ish.per)
            send.append(nq[a:b] - ish.lo)
        self.send_rows = [int(x.shape[0]) for x in send]
        flat = np.concatenate(send) if send else np.zeros(0, np.int64)
        self.n_need, self.n_send, self.per_i = int(need.shape[0]), int(flat.shape[0]), ish.per
        # the job-wide maximum of n_need (every rank has all the lists): what rank-agreed decisions are sized from
        # (ShardedMMSSL.n_chunks: the chunk count fixes the number AND the width of every rank's exchanges)
        self.n_need_max = max(int(x.shape[0]) for x in lists)
        self.group, self.bk = group, bk
        self.send_idx = torch.from_numpy(flat.astype(np.int64)).to(device)
        sel = sp.csr_matrix((np.ones(flat.shape[0], np.float32), (flat, np.arange(flat.shape[0]))),
                            shape=(ish.per, max(flat.shape[0], 1)))
        self.sel = bk.make_graph(sel)
        self.bytes_fraction = self.n_need / float(ish.n_pad)

    def _pack(self, x):
        if hasattr(self.bk, "pack_rows"):
            return self.bk.pack_rows(x, self.send_idx)
        return x[self.send_idx]

    def gather(self, x):
        send = self._pack(x)
        out = torch.empty((self.n_need, x.shape[1]), dtype=x.dtype, device=x.device)
        _all_to_all_rows(out, send, self.recv_rows, self.send_rows, self.group)
        _log_comm("halo_gather", out)
        return out

    def reduce(self, P):
        recv = torch.empty((max(self.n_send, 1), P.shape[1]), dtype=P.dtype, device=P.device)
        if self.n_send == 0:
            recv.zero_()
        _all_to_all_rows(recv[:self.n_send], P, self.send_rows, self.recv_rows, self.group)
        _log_comm("halo_reduce", P)
        return self.bk.spmm_raw(self.sel, False, recv, self.bk.EPI_NONE)

    def partial(self, rows, width):    # (the halo's partial products go through an all-to-all: no window)
        return None

    def gather_pair(self, a, b):       # (no grouped form: two launches)
        return self.gather(a), self.gather(b)

    def reduce_pair(self, fa, fb):
        return self.reduce(fa), self.reduce(fb)


class _ShardedItemSide(torch.autograd.Function):
    """The packed hot node over USER-ROW BLOCKS ONLY ("item-side" scheme). Rank r holds the edges of its users twice:
    `ui` = A_ui[U_r, :] ([per_u, I_pad], global item columns) and `iuT` = A_iu[:, U_r] ([I_pad, per_u]). Then

        u_r   = A_ui[U_r, :] . all_gather(i_r)                  gather of item-table size (north_star's all-gather
                                                                  before every propagation layer)
        i_r   = reduce_scatter( A_iu[:, U_r] . u_r )             local partial products over ALL items, summed across ranks
        backward: g(u_r) = A_iu[:, U_r]^T . all_gather(g(i_r)),  g(i_r) = reduce_scatter( A_ui[U_r, :]^T . g(u_r) )

    every collective moves ITEM-table bytes (the user table, twice as large for every shape of BASELINE.json, never
    travels): -33 % bytes per step against gathering both tables. The packed modal chain X -> MU -> MI goes the same way;
    its four collectives ride on GCN collectives of the same kind as grouped pairs (one RCCL launch each):
    4 L + 2 launches per step like the gather-both node.

    Column chunks (`nc`): every collective and the products on either side of it are cut into nc column chunks on nc
    lanes (see _Lanes). User-side tables stay row-major [per_u, w]; a lane's products read / write its column chunk in
    place (row-pitched operands, mmssl_spmm_ld_f32). Item-side chunks are contiguous [per_i, w / nc] buffers - what a
    reduce-scatter delivers and an all-gather takes - and are put side by side once per layer for the fuse kernel.
    The last layer's softmax needs whole rows: a launch of its own where the lanes meet (same arithmetic as the SpMM's
    fused epilogue)."""

    @staticmethod
    def forward(ctx, nm, scale, keep, ui, iuT, n_layers, r, bk, group, nc, xch, repl, u0, i0, *flat):
        # repl: the features are replicated - Fs hold ALL item rows, X is whole on every rank and never travels (see
        # ShardedMMSSL.__init__); `keep` then covers all item rows too
        Fs, Ws, bs = flat[:nm], flat[nm:2 * nm], flat[2 * nm:3 * nm]
        draw_p, ext_tick = 0.0, False
        if isinstance(keep, tuple):                  # ("draw", p, external_tick): fresh masks from the device generator
            _, draw_p, ext_tick = keep
            keep = None
        g = group
        per_u, per_i, d = u0.shape[0], i0.shape[0], u0.shape[1]
        wm = nm * d
        if xch is None:
            xch = _TableExchange(g, per_i)
        twin = (lambda p, k: p.twin(k)) if hasattr(ui, "twin") else (lambda p, k: p)
        new = lambda rows, w: torch.empty((rows, w), dtype=torch.float32, device=u0.device)      # noqa: E731
        # item-side: the projection / modal chain (the critical one) on the current stream + nc - 1 lanes, the GCN chain on nc
        # side lanes. halo: its exchanges are all-to-alls, which a capture only takes from the ORIGIN stream - so there the
        # GCN chain (lane 0) and every exchange live on the current stream and the modal chain on ONE forked stream.
        swap = isinstance(xch, HaloPlan)
        if repl and swap:
            raise RuntimeError("replicate_feats runs with the whole-table exchange (scheme 'item-side'), not the halo plan")
        G = _Lanes(bk, u0, nc, first_is_current=swap, via_main=swap)
        M = _Lanes(bk, u0, nc, base=nc, first_is_current=not swap, one_stream=swap)
        # Every table the LANES write is allocated here, on the origin stream BEFORE the fork: the caching allocator may hand
        # out a block whose last user is an earlier kernel of the allocating stream, which is only safe for writers ordered
        # behind that stream's work at allocation time - the lanes are, through the fork, and only then (a table allocated
        # mid-way would land in memory a still-running origin-stream kernel uses: seen as a wrong weight gradient).
        u_new = [new(per_u, d) for _ in range(n_layers)]
        MU = new(per_u, wm)
        G.fork()
        if swap:
            M.fork()
        with M.on(0):
            X, keep = bk.proj_forward(list(Fs), list(Ws), list(bs), keep, scale, draw_p, ext_tick)      # [per_i, nm d]
        if not swap:
            M.fork()
        elif G.cuda:
            X.record_stream(G.main)
        for t_ in [MU, X] + u_new:
            M.uses(t_)
            G.uses(t_)
        MU_v, X_v, MI_c = _chunks_of(MU, nc), _chunks_of(X, nc), [None] * nc
        us, its = [u0], [i0]
        i_c = [i0]
        if nc > 1:                                # a lane's chunk copy runs on the lane (i0 exists before the fork)
            i_c = []
            for c, v in enumerate(_chunks_of(i0, nc)):
                with G.on(c):
                    i_c.append(v.contiguous())
        # the modal chain's gather rides on the GCN gather of layer pair_at[0], its reduce-scatter on the GCN
        # reduce-scatter of layer pair_at[1]: late, because RCCL runs a group's collectives in issue order and a gather that
        # waits for the projection GEMM must not sit in front of the first GCN layers' collectives
        pair_at = (max(0, n_layers - 2), n_layers - 1)
        for l in range(n_layers):
            last = l == n_layers - 1
            u = u_new[l]
            u_v = _chunks_of(u, nc)
            if l == pair_at[0]:
                # grouped pairs are issued from the ORIGIN stream, all of them before the products they feed (a grouped
                # launch recorded on a forked stream of a hipGraph capture crashes hipStreamEndCapture, ROCm 7.2): lane c's
                # products wait for pair c only
                fulls = []
                for c in range(nc):
                    G.to_main(c)
                    M.to_main(c)
                    if repl:        # X is whole here already: only the GCN table travels (X_v[c]: a column-chunk view)
                        fulls.append((xch.gather(i_c[c]), X_v[c], G.mark()))
                    else:
                        fulls.append(tuple(xch.gather_pair(i_c[c], X_v[c].contiguous() if nc > 1 else X)) + (G.mark(),))
                for c, (i_full, X_full, ev) in enumerate(fulls):
                    G.after(c, ev)
                    M.after(c, ev)
                    with G.on(c):
                        G.uses(i_full, [c])
                        bk.spmm_raw(twin(ui, 2 + c), False, i_full, bk.EPI_NONE, out=u_v[c])
                    with M.on(c):
                        M.uses(X_full, [c])
                        bk.spmm_raw(twin(ui, 10 + c), False, X_full, bk.EPI_NONE, out=MU_v[c])
                del fulls
            else:
                for c in range(nc):             # item rows -> user rows: gather, product into the lane's column chunk
                    i_full = G.coll(c, lambda: xch.gather(i_c[c]), reads=(i_c[c],))
                    with G.on(c):
                        bk.spmm_raw(twin(ui, 2 + c), False, i_full, bk.EPI_NONE, out=u_v[c])
            if last:
                G.meet()
                with G.on(0):
                    bk.softmax_rows_(u)
                G.part()
            i_n = [None] * nc
            if l == pair_at[1]:                 # user rows -> ALL item rows (partial), summed into the owners' rows
                parts = []
                for c in range(nc):
                    with M.on(c):
                        PM = bk.spmm_raw(twin(iuT, 10 + c), False, MU_v[c], bk.EPI_NONE,
                                         out=xch.partial(iuT.shape[0], MU_v[c].shape[1]))
                    with G.on(c):
                        P = bk.spmm_raw(twin(iuT, 2 + c), False, u_v[c], bk.EPI_NONE,
                                        out=xch.partial(iuT.shape[0], u_v[c].shape[1]))
                    parts.append((P, PM))
                evs = []
                for c, (P, PM) in enumerate(parts):
                    G.to_main(c)
                    M.to_main(c)
                    i_n[c], MI_c[c] = xch.reduce_pair(P, PM)
                    G.uses(i_n[c], [c])
                    M.uses(MI_c[c], [c])
                    for t_ in (P, PM):          # allocated on a lane, read by the collective issued from the origin stream
                        if G.cuda:
                            t_.record_stream(G.main)
                    evs.append(G.mark())
                for c, ev in enumerate(evs):
                    G.after(c, ev)
                    M.after(c, ev)
                del parts
            else:
                for c in range(nc):
                    with G.on(c):
                        P = bk.spmm_raw(twin(iuT, 2 + c), False, u_v[c], bk.EPI_NONE,
                                        out=xch.partial(iuT.shape[0], u_v[c].shape[1]))
                    i_n[c] = G.coll(c, lambda: xch.reduce(P), reads=(P,))
            if nc > 1 or last:
                G.meet()
            with G.on(0):
                i = torch.cat(i_n, 1) if nc > 1 else i_n[0]
                if last:
                    bk.softmax_rows_(i)
            i_c = i_n
            us.append(u)
            its.append(i)
        M.meet()
        with M.on(0):
            MI = torch.cat(MI_c, 1) if nc > 1 else MI_c[0]
        M.join()
        G.join()
        if G.cuda:
            for t_ in us[1:] + its[1:] + [MI]:
                t_.record_stream(G.main)
        inv = 1.0 / (n_layers + 1)
        u_g, i_g, ss = bk.fuse_fwd(us, MU, its, MI, inv, nm, r)
        ctx.save_for_backward(MU, MI, us[-1], its[-1], keep, *Fs)
        ctx.cfg = (nm, float(scale), ui, iuT, n_layers, float(r), inv, bk, g, [b is not None for b in bs], nc, xch, bool(repl))
        ctx.set_materialize_grads(False)
        return u_g, i_g, ss, MI, MU

    @staticmethod
    def backward(ctx, Gu, Gi, g_ss, G_MI, G_MU):
        MU, MI, uG, iG, keep = ctx.saved_tensors[:5]
        Fs = ctx.saved_tensors[5:]
        nm, scale, ui, iuT, n_layers, r, inv, bk, g, has_b, nc, xch, repl = ctx.cfg
        per_u, per_i, d = MU.shape[0], MI.shape[0], uG.shape[1]
        wm = nm * d
        Gu = Gu.contiguous() if Gu is not None else torch.zeros_like(uG)
        Gi = Gi.contiguous() if Gi is not None else torch.zeros_like(iG)
        g_ss = g_ss.contiguous().to(torch.float32) if g_ss is not None else None
        G_MI = G_MI.contiguous() if G_MI is not None else None
        G_MU = G_MU.contiguous() if G_MU is not None else None
        twin = (lambda p, k: p.twin(k)) if hasattr(ui, "twin") else (lambda p, k: p)
        new = lambda rows, w: torch.empty((rows, w), dtype=torch.float32, device=Gu.device)      # noqa: E731
        swap = isinstance(xch, HaloPlan)           # (see forward)
        G = _Lanes(bk, Gu, nc, first_is_current=swap, via_main=swap)
        M = _Lanes(bk, Gu, nc, base=nc, first_is_current=not swap, one_stream=swap)
        gu_new = [new(per_u, d) for _ in range(n_layers)]          # allocated before the fork: see forward
        t = new(per_u, wm)
        G.fork()
        gMU, g_u0, gMI = bk.fuse_bwd(MU, Gu, G_MU, MI, Gi, G_MI, nm, r, inv, g_ss)         # current stream
        if G.cuda:
            # from here on the gradient of u_0 exists; the gradient of i_0 completes on GCN lane 0: a step object may
            # update the (sharded, never all-reduced) embedding tables there while the weight gradient still runs
            bk.after_fuse_bwd = G.main.record_event()
            bk.tables_stream = G.streams[0]
            for t_ in (uG, iG, Gu, Gi, g_u0):
                G.uses(t_)
        M.fork()
        for t_ in [t, gMU, gMI] + gu_new:
            M.uses(t_)
            G.uses(t_)
        t_v, gMU_v, gMI_v = _chunks_of(t, nc), _chunks_of(gMU, nc), _chunks_of(gMI, nc)
        with G.on(0):
            gi = bk.softmax_rows_bwd(iG, Gi, inv)                   # g before the item-side softmax, [per_i, d]
            gi_c = _contig_chunks(gi, nc) if nc > 1 else [gi]
        for c in range(1, nc):
            G.uses(gi_c[c], [c])
        G.part()
        gX_c = [None] * nc
        for l in range(n_layers, 0, -1):
            first = l == n_layers               # the last layer comes first; the modal chain's collectives ride on its pair
            gu = gu_new[l - 1]
            gu_v, Gu_v = _chunks_of(gu, nc), _chunks_of(Gu, nc)
            if first:                           # g(u_l) = inv Gu + A_iu[:, U_r]^T . all_gather(g(i_l))
                fulls = []
                for c in range(nc):             # (grouped pairs from the origin stream, see forward)
                    G.to_main(c)
                    M.to_main(c)
                    fulls.append(tuple(xch.gather_pair(gi_c[c], gMI_v[c].contiguous() if nc > 1 else gMI)) + (G.mark(),))
                for c, (gP, gPM, ev) in enumerate(fulls):
                    G.after(c, ev)
                    M.after(c, ev)
                    with G.on(c):
                        G.uses(gP, [c])
                        bk.spmm_raw(twin(iuT, 2 + c), True, gP, bk.EPI_AXPY, Gu_v[c], inv, out=gu_v[c])
                    with M.on(c):           # t = g(MU) = own branch + A_iu[:, U_r]^T . all_gather(g(MI))
                        M.uses(gPM, [c])
                        bk.spmm_raw(twin(iuT, 10 + c), True, gPM, bk.EPI_AXPY, gMU_v[c], 1.0, out=t_v[c])
                del fulls
            else:
                for c in range(nc):
                    gP = G.coll(c, lambda: xch.gather(gi_c[c]), reads=(gi_c[c],))
                    with G.on(c):
                        bk.spmm_raw(twin(iuT, 2 + c), True, gP, bk.EPI_AXPY, Gu_v[c], inv, out=gu_v[c])
            if first:
                G.meet()
                with G.on(0):
                    gu = bk.softmax_rows_bwd(uG, gu, 1.0)
                    gu_v = _chunks_of(gu, nc)
                G.uses(gu)
                G.part()
            gi_n = [None] * nc
            Gi_v = _chunks_of(Gi, nc)
            if first:                           # g(i_{l-1}) = inv Gi + reduce_scatter( A_ui[U_r, :]^T . g(u_l) )
                parts = []
                for c in range(nc):
                    with M.on(c):           # g(X) = dropout-backward( reduce_scatter( A_ui[U_r, :]^T . t ) )
                        part_m = bk.spmm_raw(twin(ui, 10 + c), True, t_v[c], bk.EPI_NONE,
                                             out=xch.partial(ui.shape[1], t_v[c].shape[1]))
                    with G.on(c):
                        part_g = bk.spmm_raw(twin(ui, 2 + c), True, gu_v[c], bk.EPI_NONE,
                                             out=xch.partial(ui.shape[1], gu_v[c].shape[1]))
                    parts.append((part_g, part_m))
                rgs = []
                for c, (part_g, part_m) in enumerate(parts):
                    G.to_main(c)
                    M.to_main(c)
                    if repl:        # g(X) stays a PARTIAL over this rank's users, whole rows: the weight gradient formed from
                        rg, gX_c[c] = xch.reduce(part_g), part_m      # it is summed by the replicated parameters' all-reduce
                    else:
                        rg, gX_c[c] = xch.reduce_pair(part_g, part_m)
                    G.uses(rg, [c])
                    M.uses(gX_c[c], [c])
                    for t_ in (part_g, part_m):
                        if G.cuda:
                            t_.record_stream(G.main)
                    rgs.append((rg, G.mark()))
                for c, (rg, ev) in enumerate(rgs):
                    G.after(c, ev)
                    M.after(c, ev)
                    with G.on(c):
                        gi_n[c] = rg.add_(Gi_v[c], alpha=inv)
                del parts, rgs
            else:
                for c in range(nc):
                    with G.on(c):
                        part_g = bk.spmm_raw(twin(ui, 2 + c), True, gu_v[c], bk.EPI_NONE,
                                             out=xch.partial(ui.shape[1], gu_v[c].shape[1]))
                    rs = G.coll(c, lambda: xch.reduce(part_g), reads=(part_g,))
                    with G.on(c):
                        gi_n[c] = rs.add_(Gi_v[c], alpha=inv)
            gi_c = gi_n
            if first:
                # ---- the weight gradient (the modal chain's stream) next to the rest of the GCN chain
                M.meet()
                with M.on(0):
                    gX = torch.cat(gX_c, 1) if nc > 1 else gX_c[0]
                    if keep is not None:
                        gX = bk.mask_packed(gX, keep, d, scale)
                    gW, gb = bk.proj_wgrad(gX, list(Fs), any(has_b))
                if swap and G.cuda:
                    for t_ in list(gW) + [x for x in (gb or []) if x is not None]:
                        t_.record_stream(G.main)
        G.meet()
        with G.on(0):
            gi0 = torch.cat(gi_c, 1) if nc > 1 else gi_c[0]
        M.join()
        G.join()
        if G.cuda:
            gi0.record_stream(G.main)
        grads_b = [(gb[k] if (gb is not None and has_b[k]) else None) for k in range(nm)]
        bk.table_grads = (g_u0.data_ptr(), gi0.data_ptr())
        return (None,) * 12 + (g_u0, gi0) + (None,) * nm + tuple(gW) + tuple(grads_b)


class ShardedHotPathStep:
    """forward -> BPR + 2x InfoNCE + feat-reg -> backward -> bucketed all-reduce of the replicated
    gradients -> AdamW, over row shards (the N-rank counterpart of hotpath.HotPathStep)."""

    def __init__(self, model, graphs, batch_size, n_items, group=None, lr=5.5e-4, modal_empty=False,
                 optimizer=True, fused=True, batch_rows=True):
        """batch_rows (product backend, empty modal graphs): the fused tables are computed at the batch's rows only in
        front of the loss section - all it reads - like hotpath.HotPathStep(batch_rows=True); the regulariser's sums come
        out of the backward's fuse kernel. False: the dense two-sided fuse launch."""
        self.fused = fused
        self.batch_rows = bool(batch_rows)
        self.model, self.graphs, self.group = model, tuple(graphs), group
        self.batch_size, self.n_items, self.modal_empty = int(batch_size), int(n_items), modal_empty
        dev = model.E_u.device
        self.batch = torch.zeros((3, batch_size), dtype=torch.int64, device=dev)         # users / pos / neg
        self.users, self.pos, self.neg = self.batch[0], self.batch[1], self.batch[2]
        on_gpu = dev.type == "cuda"
        if not optimizer:
            self.optimizer = None
        elif on_gpu:
            from .optim import FusedAdamW
            self.optimizer = FusedAdamW(model.parameters(), lr=lr)
        else:
            self.optimizer = torch.optim.AdamW(model.parameters(), lr=lr)
        self.loss = torch.zeros((), device=dev)
        self._loss_w = None
        self._graph = None
        # parity runs inject fixed uint8 dropout keep-masks (img, txt), each [per_items, d]; None = fresh Philox masks
        self.keep_masks = None
        self.stream = torch.cuda.Stream(device=dev) if on_gpu else None
        if on_gpu:
            self.stream.wait_stream(torch.cuda.current_stream(dev))

    def _ctx(self):
        import contextlib
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def set_batch(self, users, pos=None, neg=None):
        with self._ctx():
            if pos is None:                  # packed [3, B] batch: one device-to-device copy instead of three
                self.batch.copy_(users, non_blocking=True)
            else:
                self.users.copy_(users)
                self.pos.copy_(pos)
                self.neg.copy_(neg)

    def losses(self, keep_masks=None):
        """Returns (roots, grads, local_total, feat_local): torch.autograd.backward(roots, grads) is the step's backward;
        local_total = replicated loss terms + THIS rank's share of the regulariser."""
        m, bk, c, g = self.model, self.model.bk, self.model.cfg, self.group
        eager = self.fused and hasattr(bk, "ops")
        rows_mode = bool(eager and self.batch_rows and self.modal_empty)
        if eager:
            bk.defer_ss, bk.ss_parts = True, None
            bk.reg_parts, bk.reg_ss = None, None
            # batch-rows form: the loss reads the fused tables at the batch's rows only (global ids; the kernel keeps the
            # owned ones), see HipBackend.batch_rows
            bk.batch_rows = (self.users, self.batch[1:3].reshape(-1), m.ush.lo, m.ish.lo) if rows_mode else None
        try:
            o = m(self.graphs, keep_masks=keep_masks, modal_empty=self.modal_empty, fused=self.fused)
        finally:
            if eager:
                bk.defer_ss, bk.batch_rows = False, None
        rows_mode = rows_mode and m.last_fused and bk.reg_ss is not None
        fused = self.fused and m.last_fused
        items = self.batch[1:3].reshape(-1)           # positive then negative items: one [2B] index list, one piece
        if self.modal_empty:       # the id views are exact zeros: nothing to gather for them
            u, ia = GatherBatchRowsMulti.apply(g, 2, bk, o[0], o[1], self.users, items, m.ush.lo, m.ish.lo)
            zc = getattr(self, "_zero_rows", None)
            if zc is None or zc.shape != u.shape or zc.device != u.device:
                zc = self._zero_rows = torch.zeros_like(u)
            z_img = z_txt = zc
        else:
            u, ia, z_img, z_txt = GatherBatchRowsMulti.apply(
                g, 4, bk, o[0], o[1], o[8], o[9], self.users, items, self.users, self.users,
                m.ush.lo, m.ish.lo, m.ush.lo, m.ush.lo)
        feat_c = c.feat_reg_decay * 0.5 / self.n_items
        if fused:
            if self._loss_w is None or self._loss_w.device != u.device:
                self._loss_w = torch.tensor([1.0, 1.0, 1.0, c.cl_rate, c.cl_rate], dtype=torch.float32, device=u.device)
                self._feat_c = torch.full((), feat_c, dtype=torch.float32, device=u.device)
            ss = m._feat_ss_local
            if hasattr(bk, "ops"):
                # product backend: the loss section as ONE chain of launches rooted at the terms' known gradients (the
                # loss weights); its last launch assembles replicated terms + c * local regulariser into self.loss and
                # advances the step-owned counters (see hotpath.HotPathStep._losses_eager)
                import types
                holder = types.SimpleNamespace(ss_parts=bk.ss_parts, prefill_buf=None)      # the node's unreduced |.|^2
                bk.ss_parts = None
                terms = bk.batch_losses_rows(u, ia, z_img, z_txt, c.decay, self.batch_size, c.tau, eager_w=self._loss_w,
                                             tail=(None if rows_mode else ss.detach(), feat_c, self.loss,
                                                   getattr(self, "_ticks", None)), hot=holder)
                if rows_mode:       # the regulariser joins the loss after the backward (its sums come out of the fuse backward)
                    return [terms, ss], [self._loss_w, self._feat_c], self.loss, None
                return [terms, ss], [self._loss_w, self._feat_c], self.loss, (feat_c * ss).detach()
            terms = bk.batch_losses_rows(u, ia, z_img, z_txt, c.decay, self.batch_size, c.tau)
            total_local = bk.loss_assemble(terms, self._loss_w, ss, feat_c)
            return [total_local], [None], total_local.detach(), (feat_c * ss).detach()
        B = self.batch_size
        mf, emb = bk.bpr(u, ia[:B], ia[B:], c.decay, self.batch_size)
        feat_local = c.feat_reg_decay * ((0.5 * bk.sumsq(o[2]) + 0.5 * bk.sumsq(o[3]) + 0.5 * bk.sumsq(o[4])
                                          + 0.5 * bk.sumsq(o[5])) / self.n_items)
        cl1 = bk.infonce(z_img, u, c.tau)
        cl2 = bk.infonce(z_txt, u, c.tau)
        total_local = mf + emb + c.cl_rate * (cl1 + cl2) + feat_local
        return [total_local], [None], total_local.detach(), feat_local.detach()

    def backward(self, keep_masks=None):
        roots, grads, local_total, feat_local = self.losses(keep_masks if keep_masks is not None else self.keep_masks)
        for p in self.model.parameters():
            p.grad = None
        torch.autograd.backward(roots, grads)
        if feat_local is None:
            # batch-rows form: loss += c * sum |Mod|^2 (this rank's rows) from the backward fuse kernel's partials
            bk_, c_ = self.model.bk, self.model.cfg
            feat_c = c_.feat_reg_decay * 0.5 / self.n_items
            if bk_.reg_parts is None:
                raise RuntimeError("batch-rows step: the node's backward did not leave the regulariser partials (was the "
                                   "forward of another node run on this backend between losses() and backward()?)")
            bk_.ops.loss_add_partials(bk_.reg_parts, feat_c, self.loss, bk_.reg_ss)
            feat_local = None if _solo(self.group) else feat_c * bk_.reg_ss
            bk_.reg_parts, bk_.reg_ss = None, None
        side = getattr(self.model.bk, "tables_stream", None) if getattr(self, "_tables_early", False) else None
        if side is not None and self.optimizer is not None and self.model.bk.after_fuse_bwd is not None:
            m = self.model
            main = torch.cuda.current_stream(self.loss.device)
            side.wait_event(m.bk.after_fuse_bwd)
            # only ahead of the current stream if the tables' .grad ARE the node's buffers (see hotpath.HotPathStep._step)
            tg = getattr(m.bk, "table_grads", None)
            if tg is None or any(p.grad is None or p.grad.data_ptr() != g for p, g in zip((m.E_u, m.E_i), tg)):
                side.wait_event(main.record_event())
            with torch.cuda.stream(side):
                self.optimizer.step(external_tick=True, exclude=m.replicated_parameters())
            self._tables_join = (main, side)
            self._tables_joined = True
        # replicated dense parameters: partial (local-row) gradients -> ONE all-reduce of a persistent flat bucket that also
        # carries this rank's regulariser share in its last slot (no second collective for one scalar); the gradients are
        # packed by one multi-tensor copy and afterwards ARE views of the bucket (no copy back)
        params = [p for p in self.model.replicated_parameters() if p.grad is not None]
        solo = _solo(self.group)
        if solo:                         # one rank: the local values ARE the global ones (only the accounting stays)
            if COMM["log"] is not None and params:
                n = sum(p.grad.numel() for p in params) + 1
                COMM["log"].append(("all_reduce", (n,), 4 * n))
            if local_total is not self.loss:
                self.loss.copy_(local_total)
            return self.loss
        sizes = [p.grad.numel() for p in params]
        key = tuple(sizes)
        if getattr(self, "_bucket_key", None) != key:
            self._bucket = torch.empty(sum(sizes) + 1, dtype=torch.float32, device=self.loss.device)
            self._bucket_key = key
        flat = self._bucket
        views, k = [], 0
        for p, n in zip(params, sizes):
            views.append(flat[k:k + n].view_as(p.grad))
            k += n
        if params:
            torch._foreach_copy_(views, [p.grad for p in params])
        flat[-1:].copy_(feat_local.reshape(1))
        _all_reduce(flat, self.group)
        _log_comm("all_reduce", flat)
        for p, v in zip(params, views):
            p.grad = v
        total = local_total - feat_local + flat[-1]      # replicated part + GLOBAL regulariser
        self.loss.copy_(total)
        return total

    def _step(self):
        ops_ = getattr(self.model.bk, "ops", None)
        m = self.model
        packed = self.fused and m.bk.packed_supported([m.image_feats.shape[1], m.text_feats.shape[1]], m.ish.per,
                                                      m.cfg.embed_size)
        own_ticks = (ops_ is not None and self.optimizer is not None and packed
                     and hasattr(self.optimizer, "step_counter"))
        self._ticks = None
        if own_ticks:          # the loss-assembly launch advances the RNG and AdamW counters (no tick launches)
            dev = self.loss.device
            self._ticks = ([self.optimizer.step_counter(0, dev).data_ptr()], [ops_._rng_state(dev).data_ptr() + 8])
        self.model._external_ticks = bool(own_ticks)        # the mask draw leaves its counter to that launch too
        # The embedding tables are sharded: their gradients need no all-reduce and (with empty modal graphs) are complete
        # when the node's GCN chain ends - their AdamW launch runs on that chain's stream while the weight gradient and
        # the gradient bucket's all-reduce are still under way; only the replicated tensors are updated behind the
        # all-reduce on the step's own stream (same rule, same counter: the loss tail ticked it once for the step).
        bk = m.bk
        early = bool(own_ticks and self.modal_empty and hasattr(bk, "side_streams"))
        self._tables_early = early
        self._tables_joined = False
        try:
            total = self.backward()
            if self.optimizer is not None:
                if own_ticks:
                    self.optimizer.step(external_tick=True, exclude=[m.E_u, m.E_i] if self._tables_joined else None)
                else:
                    self.optimizer.step()
            tj, self._tables_join = getattr(self, "_tables_join", None), None
            if tj is not None:
                tj[0].wait_stream(tj[1])
        finally:
            self._ticks = None
            self._tables_early = False
            self.model._external_ticks = False
            # dropped HERE, not at the start of the next step: a tensor of this step released inside the next step's
            # capture region ends in a segfault inside hipStreamEndCapture (ROCm 7.2)
            bk.after_fuse_bwd, bk.table_grads, bk.tables_stream = None, None, None
        return total

    def step(self):
        with self._ctx():
            return self._step()

    def capture(self, warmup=3):
        """OPT-IN: capture the whole sharded step, RCCL collectives included, into a hipGraph (all on
        this object's stream). Returns False and stays eager if the runtime refuses."""
        try:
            with torch.cuda.stream(self.stream):
                for _ in range(warmup):
                    self._step()
            torch.cuda.synchronize()
            if dist.is_initialized() and dist.get_backend(self.group) == "nccl":
                # The process group's watchdog thread polls the completion events of the collectives issued so far (every
                # ~100 ms) until it has seen them complete. A poll that lands inside the capture fails on ROCm 7.2 with
                # hipErrorCapturedEvent ("event last recorded in a capturing stream") for a work issued on this stream
                # just before - and the exception in that thread terminates the process (seen on the 12.5 M-edge shape,
                # where a capture takes long enough to meet a poll). Everything is complete here: give the watchdog two of
                # its cycles to retire its list before the capture begins.
                import time
                time.sleep(0.3)
            g = torch.cuda.CUDAGraph()
            # thread_local: the process group's watchdog thread keeps querying the events of earlier collectives
            # while this thread captures; in the default (global) mode that query is an illegal call during capture
            # and aborts the process ("operation not permitted when stream is capturing")
            with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
                self._step()
            torch.cuda.synchronize()
            self._graph = g
            return True
        except Exception as e:       # pragma: no cover
            self._graph, self.capture_error = None, repr(e)
            torch.cuda.synchronize()
            return False

    def run(self):
        with self._ctx():
            if self._graph is not None:
                self._graph.replay()
            else:
                self._step()


# ---------------------------------------------------------------------------------------------
# bench.py helper (N > 1)
# ---------------------------------------------------------------------------------------------
def spawn_rank_probe(argv, port_offset=1, timeout=180):
    """Run `argv` as a child of THIS rank with the launcher's RANK/LOCAL_RANK/WORLD_SIZE but a private
    rendezvous (MASTER_PORT + port_offset, hosted by the rank-0 child instead of the launcher's agent
    store), so that all ranks' children form their own process group. Used to try something that can
    abort or hang a process (hipGraph capture with RCCL collectives inside) without losing the parent.
    Returns True iff this rank's child exited 0 within `timeout` seconds; callers must still agree on
    the answer across ranks (all-reduce MIN)."""
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + port_offset)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
    try:
        r = subprocess.run(argv, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout)
        return r.returncode == 0
    except Exception:
        return False


def comm_replay_ms(log, group, dev, iters=10, halo=None):
    """Time the collectives of one step ALONE (same kinds and sizes, dummy buffers, back to back on one stream):
    the per-step communication time if nothing overlapped. HIP events on the issuing stream. Halo exchanges are replayed
    through their plan (pack + all-to-all / all-to-all + selection SpMM)."""
    world = dist.get_world_size(group)
    bufs = []
    for kind, shape, _ in log:
        if kind in ("halo_gather", "halo_reduce"):
            rows = halo.per_i if kind == "halo_gather" else halo.n_need
            bufs.append((kind, [(torch.zeros((rows, shape[1]), dtype=torch.float32, device=dev), None)]))
            continue
        shapes = shape if (len(shape) and isinstance(shape[0], tuple)) else (shape,)      # grouped pair: several tensors
        members = []
        for sh in shapes:
            full = torch.zeros(sh, dtype=torch.float32, device=dev)
            per = (sh[0] // world,) + tuple(sh[1:]) if kind != "all_reduce" else sh
            members.append((full, torch.zeros(per, dtype=torch.float32, device=dev)))
        bufs.append((kind, members))

    def once():
        import contextlib
        for kind, members in bufs:
            cm = _coalesced(group, dev)[0] if len(members) > 1 else contextlib.nullcontext()
            with cm:
                for full, part in members:
                    if kind == "halo_gather":
                        halo.gather(full)
                    elif kind == "halo_reduce":
                        halo.reduce(full)
                    elif kind == "all_gather":
                        _all_gather_into(full, part, group)
                    elif kind == "reduce_scatter":
                        if dist.get_backend(group) == "gloo":
                            dist.all_reduce(full, group=group)
                        else:
                            dist.reduce_scatter_tensor(part, full, group=group)
                    else:
                        dist.all_reduce(full, group=group)
    once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        once()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _exchange_edges_to_item_owners(users_g, items, ish, world, rank, dev):
    """Every rank holds the edges of ITS users (global ids); the owner of item block I_r needs all edges whose
    item falls into I_r. One variable-size all-to-all of (user, item) pairs over the job's process group."""
    owner = items // ish.per
    order = np.argsort(owner, kind="stable")
    send = torch.from_numpy(np.stack((users_g[order], items[order]), 1).astype(np.int64)).to(dev)
    counts = torch.from_numpy(np.bincount(owner, minlength=world).astype(np.int64)).to(dev)
    recv_counts = torch.empty_like(counts)
    dist.all_to_all_single(recv_counts, counts)
    rc = recv_counts.cpu().tolist()
    recv = torch.empty((int(sum(rc)), 2), dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv, send, output_split_sizes=rc, input_split_sizes=counts.cpu().tolist())
    recv = recv.cpu().numpy()
    return recv[:, 0], recv[:, 1]


def build_sharded_graph(a, rank, world, dev, scaling, scheme="gather-both"):
    """The job's graph as this rank's two local matrices (global column / row ids):
        scheme gather-both : A_ui[U_r, :], A_iu[I_r, :]   (CSR row blocks of both directions)
        scheme item-side   : A_ui[U_r, :], A_iu[:, U_r]   (this rank's users' edges only, seen from both sides)

    strong : the `a.workload` shape itself (configs[3]: Amazon-Baby sharded N-way). The global graph is small
             (0.2 s to generate), so every rank generates it from the same seed and keeps its part.
    weak   : the shape x world. Each rank generates ONLY the interactions of its own user block (per-rank seed).
             gather-both: the (user, item) pairs are routed to the item owners by one all-to-all; item-side: no edge leaves
             its rank - only the items' global degrees (A_iu's 1/sqrt(deg) values) are summed by one all-reduce. No rank
             ever holds the global edge list. `synth` (configs[4]) is defined for the whole 8-GPU job, so its per-rank
             share is 2M/8 users x 1M/8 items x 100M/8 edges whatever N is.
    Returns (ui_local, iu_local, ush, ish, U, I, E_global, dv, dt)."""
    from . import synth
    U, I, E, dv, dt = synth.SHAPES[a.workload]
    if a.workload == "synth-full":
        # configs[4] WHOLE on one rank: the vertical stack of the 8 user blocks the 8-rank `synth` job generates (same seeds),
        # i.e. the very graph that job trains on - the N = 1 denominator of its speed-up
        if world != 1:
            raise ValueError("workload 'synth-full' is the one-GPU form of configs[4]; with N ranks use --workload synth")
        raw = sp.vstack(synth.stress_blocks(8)).tocsr()
        ui, iu = synth.normalised_pair(raw)
        return ui, iu, RowShard(U, 1, 0), RowShard(I, 1, 0), U, I, int(raw.nnz), dv, dt
    if a.workload == "synth":
        U, I, E, dv, dt = U // 8, I // 8, E // 8, 128, 128
        scaling = "weak"
    if scaling == "strong":
        raw = synth.interaction_matrix(U, I, E, seed=1)
        ui, iu = synth.normalised_pair(raw)
        ush, ish = RowShard(U, world, rank), RowShard(I, world, rank)
        iu_l = shard_graph_cols(iu, ish, ush) if scheme == "item-side" else shard_graph(iu, ish, ush)
        return shard_graph(ui, ush, ish), iu_l, ush, ish, U, I, int(raw.nnz), dv, dt
    Ug, Ig = U * world, I * world
    ush, ish = RowShard(Ug, world, rank), RowShard(Ig, world, rank)
    # my users' interactions with ALL items (item popularity shared by all ranks through the seed of `p`)
    raw_u = synth.interaction_matrix(U, Ig, E, seed=1000 + rank, item_seed=77)
    ui_l = synth.normalised_rows(raw_u)                               # [U, Ig] rows of A_ui
    ui_l.resize((ush.per, ish.n_pad))
    t = torch.tensor([raw_u.nnz], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(t)
    if scheme == "item-side":
        deg = torch.from_numpy(np.asarray(raw_u.sum(0)).ravel().astype(np.int64)).to(dev)      # my users' share of every item's degree
        if world > 1:
            dist.all_reduce(deg)
        sc = np.power(deg.cpu().numpy().astype(np.float64) + 1e-8, -0.5)
        iu_l = (sp.diags(sc) @ raw_u.T.tocsr().astype(np.float32)).tocsr().astype(np.float32)  # A_iu[:, U_r]: [Ig, U]
        iu_l.resize((ish.n_pad, ush.per))
        return ui_l, iu_l, ush, ish, Ug, Ig, int(t.item()), dv, dt
    coo = raw_u.tocoo()
    if world > 1:
        ug, it = _exchange_edges_to_item_owners(coo.row.astype(np.int64) + ush.lo, coo.col.astype(np.int64), ish,
                                                world, rank, dev)
    else:
        ug, it = coo.row.astype(np.int64), coo.col.astype(np.int64)
    raw_i = sp.csr_matrix((np.ones(ug.shape[0], np.float32), (it - ish.lo, ug)), shape=(ish.per, ush.n_pad))
    iu_l = synth.normalised_rows(raw_i)                               # [I_per, Ug] rows of A_iu
    return ui_l, iu_l, ush, ish, Ug, Ig, int(t.item()), dv, dt


def _bench_shared_weights(a, dv, dt):
    """The replicated parameters of the bench model (same on every rank): image / text projection, fusion weights."""
    g = torch.Generator().manual_seed(2022)

    def xavier(rows, cols):
        bound = (6.0 / (rows + cols)) ** 0.5
        return (torch.rand(rows, cols, generator=g) * 2 - 1) * bound
    return xavier(a.d, dv), xavier(a.d, dt), xavier(4 * a.d, a.d)


def _bench_feat_block(q, per_i, dv, dt):
    gq = torch.Generator().manual_seed(7 + 1000 * q)
    return torch.randn(per_i, dv, generator=gq), torch.randn(per_i, dt, generator=gq)


def _bench_table_block(a, q, per_u, per_i, U, I):
    ge = torch.Generator().manual_seed(99 + q)
    e_u = (torch.rand(per_u, a.d, generator=ge) * 2 - 1) * (6.0 / (U + a.d)) ** 0.5
    e_i = (torch.rand(per_i, a.d, generator=ge) * 2 - 1) * (6.0 / (I + a.d)) ** 0.5
    return e_u, e_i


def bench_check_masks(q, per_i, d, rows=None):
    """Injected dropout keep-masks of item-row block q for the first-step loss check (bench.py `loss_vs_n1`): a pure
    function of the block, so the sharded job and its one-GPU form drop the same entries."""
    g = torch.Generator().manual_seed(555 + q)
    return [(torch.rand(per_i if rows is None else rows, d, generator=g) >= 0.2).to(torch.uint8) for _ in range(2)]


def _bench_model(a, bk, cfg, ush, ish, group, scheme, repl, need, dev, dv, dt, img_l, txt_l, e_u, e_i):
    model = ShardedMMSSL.__new__(ShardedMMSSL)
    nn.Module.__init__(model)
    model.bk, model.cfg, model.ush, model.ish, model.group = bk, cfg, ush, ish, group
    model.scheme, model.chunks = scheme, int(getattr(a, "chunks", 0))
    model.replicate_feats = repl
    if scheme == "halo":
        model.halo = HaloPlan(need, ish, None, bk, dev)
    img_w, txt_w, w_cat = _bench_shared_weights(a, dv, dt)
    model.img_w = nn.Parameter(img_w)
    model.img_b = nn.Parameter(torch.zeros(a.d))
    model.txt_w = nn.Parameter(txt_w)
    model.txt_b = nn.Parameter(torch.zeros(a.d))
    model.w_cat = nn.Parameter(w_cat)
    model.E_u = nn.Parameter(e_u)
    model.E_i = nn.Parameter(e_i)
    model.register_buffer("image_feats", img_l, persistent=False)
    model.register_buffer("text_feats", txt_l, persistent=False)
    return model.to(dev).train()


def build_whole_bench_step(a, world, dev, scaling, group1):
    """The SAME job build_bench_step(a, rank, world, ...) distributes over `world` ranks, whole in this process (a one-rank
    `group1`: every exchange is the identity): the graph is the stack of the ranks' user blocks, tables and features the
    concatenation of their blocks, the replicated weights the shared ones. bench.py's `loss_vs_n1` runs one step of it
    on rank 0 beside the sharded job's first step (same batch, same injected dropout masks)."""
    from . import synth
    from .config import configure, HotCfg
    configure([], embed_size=a.d, weight_size=str([a.d] * a.gcn_layers), batch_size=a.batch, drop_rate=0.2)
    U, I, E, dv, dt = synth.SHAPES[a.workload]
    if a.workload == "synth":
        U, I, E, dv, dt, scaling = U // 8, I // 8, E // 8, 128, 128, "weak"
    if scaling == "strong":
        raw = synth.interaction_matrix(U, I, E, seed=1)
        Ug, Ig = U, I
        shards = [(RowShard(U, world, q), RowShard(I, world, q)) for q in range(world)]
    else:
        Ug, Ig = U * world, I * world
        raw = sp.vstack([synth.interaction_matrix(U, Ig, E, seed=1000 + q, item_seed=77) for q in range(world)]).tocsr()
        shards = [(RowShard(Ug, world, q), RowShard(Ig, world, q)) for q in range(world)]
    ui, iu = synth.normalised_pair(raw)
    ush, ish = RowShard(Ug, 1, 0), RowShard(Ig, 1, 0)
    bk = HipBackend()
    with torch.cuda.device(dev):
        plans = [bk.make_graph(ui), bk.make_graph(iu)]
        e_ui = bk.make_graph(sp.csr_matrix((Ug, Ig), dtype=np.float32))
        e_iu = bk.make_graph(sp.csr_matrix((Ig, Ug), dtype=np.float32))
    tabs = [_bench_table_block(a, q, us.per, is_.per, Ug, Ig) for q, (us, is_) in enumerate(shards)]
    feats = [_bench_feat_block(q, is_.per, dv, dt) for q, (_, is_) in enumerate(shards)]
    masks = [bench_check_masks(q, is_.per, a.d) for q, (_, is_) in enumerate(shards)]
    e_u = torch.cat([t[0] for t in tabs])[:Ug]
    e_i = torch.cat([t[1] for t in tabs])[:Ig]
    img = torch.cat([f[0] for f in feats])[:Ig]
    txt = torch.cat([f[1] for f in feats])[:Ig]
    keep = tuple(torch.cat([m[k] for m in masks])[:Ig].to(dev) for k in range(2))
    model = _bench_model(a, bk, HotCfg(), ush, ish, group1, "item-side", False, None, dev, dv, dt, img, txt, e_u, e_i)
    step = ShardedHotPathStep(model, (plans[0], plans[1], e_ui, e_iu, e_ui, e_iu), a.batch, Ig, group=group1, modal_empty=True,
                              optimizer=False)
    step.keep_masks = keep
    return step


def build_bench_step(a, rank, world, dev, scaling="weak", group=None, pre_step=None):
    """bench.py workload for the sharded path. Returns (step, local_mats, plans, stats). `group`: the process group of the
    `world` ranks (None = the default one); `pre_step(step)`: called once on the freshly built step, before its first
    optimiser step (bench.py's first-step loss check)."""
    from . import synth
    from .config import configure, HotCfg
    configure([], embed_size=a.d, weight_size=str([a.d] * a.gcn_layers), batch_size=a.batch, drop_rate=0.2)
    scheme = getattr(a, "scheme", "item-side")
    if scheme == "halo" and _solo(group):
        scheme = "item-side"        # one rank, nothing exchanged: the compact item columns would only renumber the table
    ui_l, iu_l, ush, ish, U, I, E_global, dv, dt = build_sharded_graph(a, rank, world, dev, scaling,
                                                                     "item-side" if scheme == "halo" else scheme)
    bk = HipBackend()
    need = None
    if scheme == "halo":            # the item-side pair on the item ids this rank's edges reference
        need, ui_l, iu_l = halo_graphs(ui_l, iu_l)
    with torch.cuda.device(dev):
        plans = [bk.make_graph(ui_l), bk.make_graph(iu_l)]
        e_ui = bk.make_graph(sp.csr_matrix((ush.per, ish.n_pad), dtype=np.float32))
        e_iu = bk.make_graph(sp.csr_matrix((ish.per, ush.n_pad), dtype=np.float32))
    cfg = HotCfg()
    from . import ops as _ops
    # item-side: replicate the constant features when that is cheaper than moving the projected ones (narrow features:
    # configs[4]); `--replicate-feats on|off` overrides
    rf = getattr(a, "replicate_feats", "auto")
    repl = (scheme == "item-side" and world > 1
            and (rf == "on" or (rf == "auto" and choose_replicate_feats(ish.n_pad, [dv, dt], a.d, world))))
    _ops.seed_dropout(2022 if repl else 2022 + rank, dev)    # independent masks per row shard; replicated features: ONE mask set
    # features / embeddings are generated per shard (rank-dependent seeds); replicated weights from a shared seed
    if repl:                                      # every rank's block, on every rank
        blocks = [_bench_feat_block(q, ish.per, dv, dt) for q in range(world)]
        img_l, txt_l = torch.cat([b[0] for b in blocks]), torch.cat([b[1] for b in blocks])
        del blocks
    else:
        img_l, txt_l = _bench_feat_block(rank, ish.per, dv, dt)
    e_u, e_i = _bench_table_block(a, rank, ush.per, ish.per, U, I)
    model = _bench_model(a, bk, cfg, ush, ish, group, scheme, repl, need, dev, dv, dt, img_l, txt_l, e_u, e_i)
    step = ShardedHotPathStep(model, (plans[0], plans[1], e_ui, e_iu, e_ui, e_iu), a.batch, I, group=group, modal_empty=True)
    aligned = world > 1 and _peer(group) is not None
    if aligned:
        # the ranks built their shards at different speeds (plans of 12.5 M edges take seconds); the peer exchange's waits
        # are bounded: meet on the host first, so that the first step's device-side waits measure the exchange, not the build
        dist.barrier(group=group)
    if pre_step is not None:
        pre_step(step)
    from . import ops
    ops.STATS.update(enabled=True, spmm_launches=0, edge_layers=0, spmm_bytes=0, unit_d=a.d)
    COMM["log"] = []
    if aligned:
        dist.barrier(group=group)         # (pre_step may have kept one rank busy: bench.py builds the whole job on rank 0)
    step.step()
    torch.cuda.synchronize()
    ops.STATS["enabled"] = False
    stats = dict(ops.STATS)
    stats["comm_log"], COMM["log"] = COMM["log"], None
    t = torch.tensor([int(round(stats["edge_layers"]))], dtype=torch.int64, device=dev)
    if world > 1:
        with torch.cuda.stream(step.stream):
            dist.all_reduce(t, group=group)
    torch.cuda.synchronize()
    stats["edge_layers_global"] = int(t.item())
    stats["edge_layers"] = int(round(stats["edge_layers"]))
    stats.update(n_users=U, n_items=I, n_edges=E_global, local_users=ush.per, local_items=ish.per,
                 local_edges=int(ui_l.nnz), scheme=scheme, replicate_feats=bool(repl),
                 chunks=(model.n_chunks(2) if (scheme in ("item-side", "halo") and not _solo(group)) else 1),
                 halo=(model.halo if scheme == "halo" else None),
                 halo_rows_fraction=(round(model.halo.bytes_fraction, 4) if scheme == "halo" else None))
    return step, (ui_l, iu_l), plans, stats
