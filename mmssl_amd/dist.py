"""Row-sharded multi-GPU hot path (one process per GPU, torch.distributed over RCCL/xGMI).

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
import numpy as np
import scipy.sparse as sp
import torch
import torch.distributed as dist
import torch.nn as nn


class HipBackend:
    """The product compute backend: HIP kernels via mmssl_amd.ops."""

    def __init__(self):
        from . import ops
        from .graph import GraphPlan
        self.ops = ops
        self.make_graph = GraphPlan
        self.EPI_NONE, self.EPI_SOFTMAX = ops.EPI_NONE, ops.EPI_SOFTMAX
        for name in ("spmm", "l2norm_rows", "linear", "bpr", "infonce", "sumsq"):
            setattr(self, name, getattr(ops, name))


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


# ---------------------------------------------------------------------------------------------
# collectives with autograd
# ---------------------------------------------------------------------------------------------
def _reduce_scatter_sum(full, per, group):
    if dist.get_backend(group) == "gloo":      # gloo has no reduce_scatter: all-reduce + slice
        full = full.contiguous()
        dist.all_reduce(full, group=group)
        r = dist.get_rank(group)
        return full[r * per:(r + 1) * per].clone()
    out = torch.empty((per,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
    dist.reduce_scatter_tensor(out, full.contiguous(), group=group)
    return out


class AllGatherRows(torch.autograd.Function):
    """[per, d] row shard -> [world*per, d]; backward = reduce-scatter(sum) of the full gradient."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group, ctx.per = group, x.shape[0]
        world = dist.get_world_size(group)
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous(), group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        return _reduce_scatter_sum(g, ctx.per, ctx.group), None


class GatherBatchRows(torch.autograd.Function):
    """Rows table[idx] of a row-sharded table, replicated on every rank. idx is global; each
    rank contributes the rows it owns (others are exact zeros) and the buffers are summed."""

    @staticmethod
    def forward(ctx, table, idx, lo, group):
        per = table.shape[0]
        mine = ((idx >= lo) & (idx < lo + per))
        local = (idx - lo).clamp(0, per - 1)
        rows = table[local] * mine.unsqueeze(1).to(table.dtype)
        dist.all_reduce(rows, group=group)
        ctx.save_for_backward(local, mine)
        ctx.per = per
        return rows

    @staticmethod
    def backward(ctx, g):
        local, mine = ctx.saved_tensors
        out = torch.zeros((ctx.per, g.shape[1]), dtype=g.dtype, device=g.device)
        out.index_add_(0, local, g * mine.unsqueeze(1).to(g.dtype))
        return out, None, None, None


# ---------------------------------------------------------------------------------------------
# sharded model + step
# ---------------------------------------------------------------------------------------------
class ShardedMMSSL(nn.Module):
    """MMSSL.forward (reference Models.py:171-220) over row shards. Parameters:
    replicated  image_trans.{weight,bias}, text_trans.{weight,bias}, w_self_attention_cat
    sharded     user_id_embedding [per_u, d], item_id_embedding [per_i, d]."""

    def __init__(self, backend, cfg, ush, ish, state, image_feats, text_feats, group=None):
        super().__init__()
        self.bk, self.cfg, self.ush, self.ish, self.group = backend, cfg, ush, ish, group
        self.img_w = nn.Parameter(state["image_trans.weight"].clone())
        self.img_b = nn.Parameter(state["image_trans.bias"].clone())
        self.txt_w = nn.Parameter(state["text_trans.weight"].clone())
        self.txt_b = nn.Parameter(state["text_trans.bias"].clone())
        self.w_cat = nn.Parameter(state["weight_dict.w_self_attention_cat"].clone())
        self.E_u = nn.Parameter(ush.slice_rows(state["user_id_embedding.weight"]))
        self.E_i = nn.Parameter(ish.slice_rows(state["item_id_embedding.weight"]))
        self.register_buffer("image_feats", ish.slice_rows(torch.as_tensor(image_feats)), persistent=False)
        self.register_buffer("text_feats", ish.slice_rows(torch.as_tensor(text_feats)), persistent=False)

    def replicated_parameters(self):
        return [self.img_w, self.img_b, self.txt_w, self.txt_b, self.w_cat]

    def _gather(self, x):
        return AllGatherRows.apply(x, self.group)

    def _fusion(self, a, b):
        c = self.cfg
        fold = self.w_cat.view(c.head_num, c.embed_size, c.embed_size).sum(0)
        return self.bk.linear(a + b, (0.5 * fold).t().contiguous())

    def forward(self, graphs, keep_masks=None, modal_empty=False):
        """graphs = local (ui, iu, img_ui, img_iu, txt_ui, txt_iu). Returns the local rows of the
        reference's 12 outputs (0/6 and 1/7 identical)."""
        bk, c = self.bk, self.cfg
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
        x_both = self._gather(torch.cat((x_img, x_txt), 1))
        img_user = bk.spmm(ui, x_both[:, :d].contiguous())
        txt_user = bk.spmm(ui, x_both[:, d:].contiguous())
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
            i = bk.spmm(iu, self._gather(u), epi)
            u_sum = u_sum + u
            i_sum = i_sum + i
        inv = 1.0 / (c.n_ui_layers + 1)
        r = c.model_cat_rate
        u_g = bk.l2norm_rows(txt_user, bk.l2norm_rows(img_user, u_sum * inv, r), r)
        i_g = bk.l2norm_rows(txt_item, bk.l2norm_rows(img_item, i_sum * inv, r), r)
        return (u_g, i_g, img_item, txt_item, img_user, txt_user, u_g, i_g, img_uid, txt_uid, img_iid, txt_iid)


class ShardedHotPathStep:
    """forward -> BPR + 2x InfoNCE + feat-reg -> backward -> bucketed all-reduce of the replicated
    gradients -> AdamW, over row shards (the N-rank counterpart of hotpath.HotPathStep)."""

    def __init__(self, model, graphs, batch_size, n_items, group=None, lr=5.5e-4, modal_empty=False,
                 optimizer=True):
        self.model, self.graphs, self.group = model, tuple(graphs), group
        self.batch_size, self.n_items, self.modal_empty = int(batch_size), int(n_items), modal_empty
        dev = model.E_u.device
        self.users = torch.zeros(batch_size, dtype=torch.int64, device=dev)
        self.pos = torch.zeros(batch_size, dtype=torch.int64, device=dev)
        self.neg = torch.zeros(batch_size, dtype=torch.int64, device=dev)
        self.optimizer = torch.optim.AdamW(model.parameters(), lr=lr) if optimizer else None
        self.loss = torch.zeros((), device=dev)

    def set_batch(self, users, pos, neg):
        self.users.copy_(users)
        self.pos.copy_(pos)
        self.neg.copy_(neg)

    def losses(self, keep_masks=None):
        m, bk, c, g = self.model, self.model.bk, self.model.cfg, self.group
        o = m(self.graphs, keep_masks=keep_masks, modal_empty=self.modal_empty)
        rows = lambda t, idx, sh: GatherBatchRows.apply(t, idx, sh.lo, g)     # noqa: E731
        u = rows(o[0], self.users, m.ush)
        p = rows(o[1], self.pos, m.ish)
        n = rows(o[1], self.neg, m.ish)
        mf, emb = bk.bpr(u, p, n, c.decay, self.batch_size)
        feat_local = c.feat_reg_decay * ((0.5 * bk.sumsq(o[2]) + 0.5 * bk.sumsq(o[3]) + 0.5 * bk.sumsq(o[4])
                                          + 0.5 * bk.sumsq(o[5])) / self.n_items)
        cl1 = bk.infonce(rows(o[8], self.users, m.ush), u, c.tau)
        cl2 = bk.infonce(rows(o[9], self.users, m.ush), u, c.tau)
        replicated = mf + emb + c.cl_rate * (cl1 + cl2)
        return replicated, feat_local

    def backward(self, keep_masks=None):
        replicated, feat_local = self.losses(keep_masks)
        for p in self.model.parameters():
            p.grad = None
        (replicated + feat_local).backward()
        # replicated dense parameters: partial (local-row) gradients -> one bucketed all-reduce
        params = [p for p in self.model.replicated_parameters() if p.grad is not None]
        if params:
            flat = torch.cat([p.grad.reshape(-1) for p in params])
            dist.all_reduce(flat, group=self.group)
            k = 0
            for p in params:
                n = p.grad.numel()
                p.grad.copy_(flat[k:k + n].view_as(p.grad))
                k += n
        feat = feat_local.detach().clone()
        dist.all_reduce(feat, group=self.group)
        total = replicated.detach() + feat
        self.loss.copy_(total)
        return total

    def step(self):
        total = self.backward()
        if self.optimizer is not None:
            self.optimizer.step()
        return total

    def run(self):
        self.step()


# ---------------------------------------------------------------------------------------------
# bench.py helper (N > 1)
# ---------------------------------------------------------------------------------------------
def build_bench_step(a, rank, world, dev):
    """Weak-scaled bench workload: the `a.workload` shape x world, seeded identically on every
    rank; each rank keeps its row blocks only. Returns (step, raw_global, mats_local, plans, stats)."""
    from . import synth
    from .config import configure, HotCfg
    U, I, E, dv, dt = synth.SHAPES[a.workload]
    if a.workload == "synth":          # the 100M-edge stress shape is defined for the WHOLE job, d=128
        U, I, E, dv, dt = U // 8 * world, I // 8 * world, E // 8 * world, 128, 128
    else:
        U, I, E = U * world, I * world, E * world
    configure([], embed_size=a.d, weight_size=str([a.d] * a.gcn_layers), batch_size=a.batch, drop_rate=0.2)
    raw = synth.interaction_matrix(U, I, E, seed=1)
    ui, iu = synth.normalised_pair(raw)
    ush, ish = RowShard(U, world, rank), RowShard(I, world, rank)
    bk = HipBackend()
    ui_l, iu_l = shard_graph(ui, ush, ish), shard_graph(iu, ish, ush)
    with torch.cuda.device(dev):
        plans = [bk.make_graph(ui_l), bk.make_graph(iu_l)]
        e_ui = bk.make_graph(sp.csr_matrix((ush.per, ish.n_pad), dtype=np.float32))
        e_iu = bk.make_graph(sp.csr_matrix((ish.per, ush.n_pad), dtype=np.float32))
    cfg = HotCfg()
    g = torch.Generator().manual_seed(2022)

    def xavier(rows, cols):
        bound = (6.0 / (rows + cols)) ** 0.5
        return (torch.rand(rows, cols, generator=g) * 2 - 1) * bound
    # features / embeddings are generated per shard from a rank-independent stream position
    gi = torch.Generator().manual_seed(7 + 1000 * rank)
    img_l = torch.randn(ish.per, dv, generator=gi)
    txt_l = torch.randn(ish.per, dt, generator=gi)
    state = {"image_trans.weight": xavier(a.d, dv), "image_trans.bias": torch.zeros(a.d),
             "text_trans.weight": xavier(a.d, dt), "text_trans.bias": torch.zeros(a.d),
             "weight_dict.w_self_attention_cat": xavier(4 * a.d, a.d)}
    ge = torch.Generator().manual_seed(99 + rank)
    bound = (6.0 / (U + a.d)) ** 0.5
    state["user_id_embedding.weight"] = None
    model = ShardedMMSSL.__new__(ShardedMMSSL)
    nn.Module.__init__(model)
    model.bk, model.cfg, model.ush, model.ish, model.group = bk, cfg, ush, ish, None
    model.img_w = nn.Parameter(state["image_trans.weight"])
    model.img_b = nn.Parameter(state["image_trans.bias"])
    model.txt_w = nn.Parameter(state["text_trans.weight"])
    model.txt_b = nn.Parameter(state["text_trans.bias"])
    model.w_cat = nn.Parameter(state["weight_dict.w_self_attention_cat"])
    model.E_u = nn.Parameter((torch.rand(ush.per, a.d, generator=ge) * 2 - 1) * bound)
    model.E_i = nn.Parameter((torch.rand(ish.per, a.d, generator=ge) * 2 - 1) * (6.0 / (I + a.d)) ** 0.5)
    model.register_buffer("image_feats", img_l, persistent=False)
    model.register_buffer("text_feats", txt_l, persistent=False)
    model = model.to(dev).train()
    step = ShardedHotPathStep(model, (plans[0], plans[1], e_ui, e_iu, e_ui, e_iu), a.batch, I, modal_empty=True)
    from . import ops
    ops.STATS.update(enabled=True, spmm_launches=0, edge_layers=0, spmm_bytes=0)
    step.step()
    torch.cuda.synchronize()
    ops.STATS["enabled"] = False
    stats = dict(ops.STATS)
    t = torch.tensor([stats["edge_layers"]], dtype=torch.int64, device=dev)
    dist.all_reduce(t)
    stats["edge_layers_global"] = int(t.item())
    return step, raw, (ui_l, iu_l), plans, stats
