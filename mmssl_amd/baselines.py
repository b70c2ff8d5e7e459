"""The two multi-modal baselines the reference ships next to MMSSL, on this package's HIP kernels
(SURVEY.md section 8f "next #4"):

    LATTICE   /root/reference/LATTICE/codes/Models.py:32-136   (cf_model ngcf / lightgcn / mf)
    MICRO     /root/reference/MICRO/codes/Models.py:13-160     (cf_model ngcf / lightgcn / mf; + its contrastive loss;
              its --sparse 1 and --sparse 0 item graphs are the same lists here)
    NGCF, LightGCN, MF   MICRO/codes/Models.py:163-243

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
    dense 18 K x 18 K matrices for Amazon-Baby; MICRO's --sparse 1 builds the same graph as a COO tensor through
    torch_scatter): `item_adj @ h` is the ELL SpMM `ops.ell_spmm` (backward: atomics for h, one dot product per stored
    entry for the LEARNED weights); the normalisation (sym / rw / none) is a few elementwise ops over the [N, k] lists;
  * NGCF layers: SpMM, the two nn.Linear products on the projection kernels (`ops.linear`, input gradient included),
    `ops.mul` and `ops.ngcf_combine` (leaky_relu + sum + dropout + normalize in one launch each way);
  * MICRO's N x N contrastive loss: `ops.infonce(..., log_eps=0)`, the InfoNCE tile kernels."""
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


def sym_normalise(idx, val, norm_type="sym"):
    """The kNN matrix given as lists, normalised (compute_normalized_laplacian, LATTICE Models.py:18-24; MICRO
    utility/norm.py:23-55, dense and sparse branch alike): 'sym' = D^-1/2 A D^-1/2 with the ROW sums on both sides,
    'rw' = D^-1 A, 'none' = A."""
    if norm_type == "none":
        return val
    deg = val.sum(1)
    if norm_type == "rw":
        d = 1.0 / deg
        d = torch.where(torch.isinf(d), torch.zeros_like(d), d)
        return d.unsqueeze(1) * val
    if norm_type != "sym":
        raise ValueError("norm_type %r" % (norm_type,))
    d = deg.pow(-0.5)
    d = torch.where(torch.isinf(d), torch.zeros_like(d), d)
    return d.unsqueeze(1) * val * d[idx]


def lists_matmul(idx, w, h):
    """(A @ h)[i] = sum_j w[i, j] * h[idx[i, j]]: the ELL SpMM kernel (gradients for w and h)."""
    return ops.ell_spmm(idx, w.contiguous(), h.contiguous())


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
                 feat_embed_dim=64, topk=TOPK, lambda_coeff=LAMBDA, cf_model="lightgcn", norm_type="sym"):
        super().__init__()
        if cf_model not in ("ngcf", "lightgcn", "mf"):
            raise ValueError("cf_model %r" % (cf_model,))
        self.n_users, self.n_items, self.embedding_dim = n_users, n_items, embedding_dim
        self.n_ui_layers = len(weight_size)
        self.topk, self.lambda_coeff, self.cf_model, self.norm_type = int(topk), float(lambda_coeff), cf_model, norm_type
        if cf_model == "ngcf":                       # same module names as the reference: its state_dict loads
            sizes = [embedding_dim] + list(weight_size)
            self.GC_Linear_list, self.Bi_Linear_list, self.dropout_list = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
            for i in range(self.n_ui_layers):
                self.GC_Linear_list.append(nn.Linear(sizes[i], sizes[i + 1]))
                self.Bi_Linear_list.append(nn.Linear(sizes[i], sizes[i + 1]))
                self.dropout_list.append(nn.Dropout(dropout_list[i]))
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
                    out.append((idx, sym_normalise(idx, val, self.norm_type)))
                self._orig = out
        return self._orig

    def _lightgcn(self, adj, h):
        """(u_g, i_g) of the reference's cf_model branches (LATTICE Models.py:118-136)."""
        if self.cf_model == "mf":
            return self.user_embedding.weight, self.item_id_embedding.weight + ops.l2norm_rows(h)
        plan = _plan(adj)
        ego = torch.cat((self.user_embedding.weight, self.item_id_embedding.weight), dim=0)
        if self.cf_model == "ngcf":
            layers = ngcf_layers(self, plan, ego)
            acc = layers[0]
            for x in layers[1:]:
                acc = acc + x
            acc = acc * (1.0 / len(layers))
            u_g, i_g = torch.split(acc, [self.n_users, self.n_items], dim=0)
            return u_g, ops.l2norm_rows(h, i_g.contiguous(), 1.0)
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


def ngcf_layers(m, plan, ego):
    """[ego_0, norm_1, ..., norm_L] of the NGCF propagation (LATTICE Models.py:106-118, MICRO Models.py:126-139, 195-204):
    side = A . ego; ego' = dropout(leaky_relu(GC(side)) + leaky_relu(Bi(ego * side))); norm = normalize(ego')."""
    out = [ego]
    for i in range(m.n_ui_layers):
        side = ops.spmm(plan, ego)
        gc, bi = m.GC_Linear_list[i], m.Bi_Linear_list[i]
        G = ops.linear(side, gc.weight, gc.bias)
        B = ops.linear(ops.mul(ego, side), bi.weight, bi.bias)
        p = float(m.dropout_list[i].p)
        keep, scale = None, 1.0
        if m.training and p > 0.0:
            keep = ops.dropout_masks(1, G.shape[0], G.shape[1], p, G.device)[0]
            scale = 1.0 / (1.0 - p)
        ego, norm = ops.ngcf_combine(G, B, keep, scale)
        out.append(norm)
    return out


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
            val = sym_normalise(idx, torch.cat((w[0] * iv, w[1] * tv), 1))        # normalised AFTER the mix (Models.py:95-97)
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
                gs.append((idx, sym_normalise(idx, val, self.norm_type)))
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


class NGCF(nn.Module):
    """MICRO/codes/Models.py:179-217: the layers' normalised embeddings CONCATENATED (not averaged)."""

    def __init__(self, n_users, n_items, embedding_dim, weight_size, dropout_list, image_feats=None, text_feats=None):
        super().__init__()
        self.n_users, self.n_items, self.n_ui_layers = n_users, n_items, len(weight_size)
        sizes = [embedding_dim] + list(weight_size)
        self.dropout_list, self.GC_Linear_list, self.Bi_Linear_list = nn.ModuleList(), nn.ModuleList(), nn.ModuleList()
        for i in range(self.n_ui_layers):
            self.GC_Linear_list.append(nn.Linear(sizes[i], sizes[i + 1]))
            self.Bi_Linear_list.append(nn.Linear(sizes[i], sizes[i + 1]))
            self.dropout_list.append(nn.Dropout(dropout_list[i]))
        self.user_embedding = nn.Embedding(n_users, embedding_dim)
        self.item_id_embedding = nn.Embedding(n_items, embedding_dim)
        nn.init.xavier_uniform_(self.user_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)

    def forward(self, adj, build_item_graph=False):
        ego = torch.cat((self.user_embedding.weight, self.item_id_embedding.weight), dim=0)
        allv = torch.cat(ngcf_layers(self, _plan(adj), ego), dim=1)
        return tuple(torch.split(allv, [self.n_users, self.n_items], dim=0))


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
