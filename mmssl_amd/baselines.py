"""The two multi-modal baselines the reference ships next to MMSSL, on this package's HIP kernels
(SURVEY.md section 8f "next #4"):

    LATTICE   /root/reference/LATTICE/codes/Models.py:32-136   (cf_model lightgcn / mf)
    MICRO     /root/reference/MICRO/codes/Models.py:13-160     (cf_model lightgcn / mf; + its contrastive loss)

Same constructor signature and `forward(adj, build_item_graph=False)` contract as the reference classes, same
parameter names (a reference state_dict loads). What runs where:

  * LightGCN propagation over the (U+I)^2 normalised adjacency: `ops.spmm` on a GraphPlan of `adj` (torch sparse
    COO, scipy or a plan), the same CSR kernel as MMSSL's propagation;
  * modality projections `image_trs` / `text_trs` (trainable raw-feature tables, so the input gradient is needed
    too): `ops.linear` (fp32 MFMA) forward, weight and input gradients;
  * item-item kNN graphs: cosine scores by the fused tile kernel `ops.sim_rows` (d <= 128; the one-time graph of the
    RAW features, d in the thousands, uses a library GEMM at construction like the reference's cached .pt files),
    top-k by `ops.topk_rows` - nothing N x N is sorted;
  * the graphs live as [N, k] neighbour lists, never as dense N x N matrices (the reference's LATTICE keeps four
    dense 18 K x 18 K matrices for Amazon-Baby); the symmetric normalisation and the `item_adj @ h` products are
    gathers over those lists (torch elementwise ops: they carry the gradient into the learned graph);
  * MICRO's N x N contrastive loss: `ops.infonce(..., log_eps=0)`, the InfoNCE tile kernels.
The NGCF variants (per-layer dense transforms) are not built."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .graph import GraphPlan

TOPK, LAMBDA, N_LAYERS = 10, 0.9, 1          # the reference's parser defaults (topk, lambda_coeff, n_layers / layers)


def _cosine_rows(x):
    """Row-normalised copy: the row kernel for embedding widths (32 / 64 / 128 / 256), torch for the raw feature
    widths of the one-time original graphs."""
    return ops.l2norm_rows(x) if x.shape[1] in (32, 64, 128, 256) else F.normalize(x, p=2, dim=1)


def knn_lists(context, topk):
    """(idx [N, k] int64, val [N, k]) of build_sim + torch.topk (LATTICE Models.py:14-27): cosine similarity of every
    row against all rows, k best per row. `val` is differentiable w.r.t. `context` (the k selected scores are
    recomputed from the normalised rows), `idx` is not - exactly like topk's values / indices."""
    cn = _cosine_rows(context)
    with torch.no_grad():
        d = cn.shape[1]
        if d in (32, 64, 128):
            S, _ = ops.sim_rows(cn.detach(), cn.detach())
        else:                                   # raw feature width: one-time construction, library GEMM
            S = torch.mm(cn.detach(), cn.detach().t())
        idx = ops.topk_rows(S, topk)
        del S
    val = (cn.unsqueeze(1) * cn[idx]).sum(-1)
    return idx, val


def sym_normalise(idx, val):
    """D^-1/2 A D^-1/2 of the kNN matrix given as lists (compute_normalized_laplacian, LATTICE Models.py:18-24):
    rowsum over the row's entries, the same vector on both sides."""
    d = val.sum(1).pow(-0.5)
    d = torch.where(torch.isinf(d), torch.zeros_like(d), d)
    return d.unsqueeze(1) * val * d[idx]


def lists_matmul(idx, w, h):
    """(A @ h)[i] = sum_j w[i, j] * h[idx[i, j]]"""
    return (w.unsqueeze(-1) * h[idx]).sum(1)


def _plan(adj):
    if isinstance(adj, GraphPlan) or hasattr(adj, "twin"):
        return adj
    plan = getattr(adj, "_mmssl_plan", None)
    if plan is None:
        plan = GraphPlan(adj)
        try:
            adj._mmssl_plan = plan
        except Exception:
            pass
    return plan


class _Base(nn.Module):
    def __init__(self, n_users, n_items, embedding_dim, weight_size, dropout_list, image_feats, text_feats,
                 feat_embed_dim=64, topk=TOPK, lambda_coeff=LAMBDA, cf_model="lightgcn"):
        super().__init__()
        if cf_model not in ("lightgcn", "mf"):
            raise NotImplementedError("cf_model %r (the NGCF variant is not built)" % cf_model)
        self.n_users, self.n_items, self.embedding_dim = n_users, n_items, embedding_dim
        self.n_ui_layers = len(weight_size)
        self.topk, self.lambda_coeff, self.cf_model = int(topk), float(lambda_coeff), cf_model
        self.user_embedding = nn.Embedding(n_users, embedding_dim)
        self.item_id_embedding = nn.Embedding(n_items, embedding_dim)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)
        self.image_embedding = nn.Embedding.from_pretrained(torch.Tensor(image_feats), freeze=False)
        self.text_embedding = nn.Embedding.from_pretrained(torch.Tensor(text_feats), freeze=False)
        self.image_trs = nn.Linear(image_feats.shape[1], feat_embed_dim)
        self.text_trs = nn.Linear(text_feats.shape[1], feat_embed_dim)
        self._orig = None            # kNN lists of the raw features, built on first forward (needs the device)

    def _original_graphs(self):
        if self._orig is None:
            with torch.no_grad():
                out = []
                for emb in (self.image_embedding, self.text_embedding):
                    idx, val = knn_lists(emb.weight.detach(), self.topk)
                    out.append((idx, sym_normalise(idx, val)))
                self._orig = out
        return self._orig

    def _lightgcn(self, adj, h):
        """(u_g, i_g) of the reference's cf_model branches (LATTICE Models.py:118-136)."""
        if self.cf_model == "mf":
            return self.user_embedding.weight, self.item_id_embedding.weight + ops.l2norm_rows(h)
        plan = _plan(adj)
        ego = torch.cat((self.user_embedding.weight, self.item_id_embedding.weight), dim=0)
        acc = ego
        for _ in range(self.n_ui_layers):
            ego = ops.spmm(plan, ego)
            acc = acc + ego
        acc = acc * (1.0 / (self.n_ui_layers + 1))
        u_g, i_g = torch.split(acc, [self.n_users, self.n_items], dim=0)
        return u_g, ops.l2norm_rows(h, i_g.contiguous(), 1.0)

    def _project(self):
        return (ops.linear(self.image_embedding.weight, self.image_trs.weight, self.image_trs.bias),
                ops.linear(self.text_embedding.weight, self.text_trs.weight, self.text_trs.bias))


class LATTICE(_Base):
    """LATTICE/codes/Models.py:32-136: one item graph = normalised(w0 kNN(image) + w1 kNN(text)) mixed with the graph
    of the raw features; `h` = n_layers products with it; LightGCN on the interaction graph; i += normalize(h)."""

    def __init__(self, *a, n_layers=N_LAYERS, **k):
        super().__init__(*a, **k)
        self.n_layers = int(n_layers)
        self.modal_weight = nn.Parameter(torch.Tensor([0.5, 0.5]))
        self._graph = None

    def forward(self, adj, build_item_graph=False):
        image_feats, text_feats = self._project()
        (oi, ow_img), (ot, ow_txt) = self._original_graphs()
        if build_item_graph or self._graph is None:
            w = torch.softmax(self.modal_weight, dim=0)
            ii, iv = knn_lists(image_feats, self.topk)
            ti, tv = knn_lists(text_feats, self.topk)
            idx = torch.cat((ii, ti), 1)                               # weight[0] * image_adj + weight[1] * text_adj
            val = sym_normalise(idx, torch.cat((w[0] * iv, w[1] * tv), 1))
            self._graph = (idx, val, w)
        else:
            idx, val, w = self._graph
            self._graph = (idx, val.detach(), w.detach())
            idx, val, w = self._graph
        lam = self.lambda_coeff
        h = self.item_id_embedding.weight
        for _ in range(self.n_layers):
            h = (1 - lam) * lists_matmul(idx, val, h) + lam * (w[0] * lists_matmul(oi, ow_img, h)
                                                               + w[1] * lists_matmul(ot, ow_txt, h))
        return self._lightgcn(adj, h)


class MICRO(_Base):
    """MICRO/codes/Models.py:13-160: one item graph PER modality, an attention over the two propagated views, the
    fused view `h` added to the LightGCN item embeddings; forward returns (u, i, image_item, text_item, h)."""

    def __init__(self, *a, layers=N_LAYERS, **k):
        k.setdefault("feat_embed_dim", a[2] if len(a) > 2 else 64)       # MICRO projects to embed_size
        super().__init__(*a, **k)
        self.layers = int(layers)
        self.query = nn.Sequential(nn.Linear(self.embedding_dim, self.embedding_dim), nn.Tanh(),
                                   nn.Linear(self.embedding_dim, 1, bias=False))
        self.tau = 0.5
        self._graphs = None

    def batched_contrastive_loss(self, z1, z2, batch_size=4096):
        """MICRO Models.py:74-95 (no constant inside the log); the row blocking only shaped the reference's memory."""
        return ops.infonce(z1, z2, self.tau, log_eps=0.0)

    def forward(self, adj, build_item_graph=False):
        image_feats, text_feats = self._project()
        (oi, ow_img), (ot, ow_txt) = self._original_graphs()
        if build_item_graph or self._graphs is None:
            gs = []
            for feats in (image_feats, text_feats):
                idx, val = knn_lists(feats, self.topk)
                gs.append((idx, sym_normalise(idx, val)))
            self._graphs = gs
        else:
            self._graphs = [(i_, v.detach()) for i_, v in self._graphs]
        lam = self.lambda_coeff
        views = []
        for (idx, val), (o_idx, o_val) in zip(self._graphs, ((oi, ow_img), (ot, ow_txt))):
            e = self.item_id_embedding.weight
            for _ in range(self.layers):
                e = (1 - lam) * lists_matmul(idx, val, e) + lam * lists_matmul(o_idx, o_val, e)
            views.append(e)
        att = torch.softmax(torch.cat([self.query(v) for v in views], dim=-1), dim=-1)
        h = att[:, 0:1] * views[0] + att[:, 1:2] * views[1]
        u_g, i_g = self._lightgcn(adj, h)
        return u_g, i_g, views[0], views[1], h


class LightGCN(nn.Module):
    """MICRO/codes/Models.py:222-242."""

    def __init__(self, n_users, n_items, embedding_dim, weight_size, dropout_list, image_feats=None, text_feats=None):
        super().__init__()
        self.n_users, self.n_items, self.n_ui_layers = n_users, n_items, len(weight_size)
        self.user_embedding = nn.Embedding(n_users, embedding_dim)
        self.item_id_embedding = nn.Embedding(n_items, embedding_dim)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)

    def forward(self, adj, build_item_graph=False):
        plan = _plan(adj)
        ego = torch.cat((self.user_embedding.weight, self.item_id_embedding.weight), dim=0)
        acc = ego
        for _ in range(self.n_ui_layers):
            ego = ops.spmm(plan, ego)
            acc = acc + ego
        acc = acc * (1.0 / (self.n_ui_layers + 1))
        return tuple(torch.split(acc, [self.n_users, self.n_items], dim=0))


class MF(nn.Module):
    """MICRO/codes/Models.py:165-178."""

    def __init__(self, n_users, n_items, embedding_dim, weight_size, dropout_list, image_feats=None, text_feats=None):
        super().__init__()
        self.user_embedding = nn.Embedding(n_users, embedding_dim)
        self.item_embedding = nn.Embedding(n_items, embedding_dim)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_embedding.weight)

    def forward(self, adj, build_item_graph=False):
        return self.user_embedding.weight, self.item_embedding.weight
