"""`MMSSL` and `Discriminator` with the reference's constructor signatures, parameter names and
forward contract (/root/reference/MMSSL/Models.py:17-245), computed by the HIP kernels of
libmmssl_hip.so instead of PyTorch op call sites:

  projection    nn.Linear + Dropout            -> ops.linear   (fp32 MFMA, fused bias+dropout)
  propagation   torch.sparse.mm x (8L + 2G)    -> ops.spmm     (CSR plan, fused last-layer softmax)
  F.normalize   + the surrounding scaled adds  -> ops.l2norm_rows (fused alpha * norm + base)

A reference `state_dict` loads unchanged (same keys incl. the aliased encoder.* / align.*
entries and the unused image_embedding / text_embedding / batch_norm parameters). Graph
arguments may be the reference's torch sparse COO tensors or `GraphPlan`s.
"""
import torch
import torch.nn as nn

from . import ops
from .config import args
from .graph import GraphPlan


def _plan_of(g):
    """GraphPlan for a graph argument; torch sparse tensors get a plan cached on the tensor."""
    if isinstance(g, GraphPlan) or hasattr(g, "twin"):      # a plan or a plan view (GraphPlan.twin)
        return g
    plan = getattr(g, "_mmssl_plan", None)
    if plan is None:
        plan = GraphPlan(g)
        try:
            g._mmssl_plan = plan
        except Exception:
            pass
    return plan


class MMSSL(nn.Module):
    def __init__(self, n_users, n_items, embedding_dim, weight_size, dropout_list, image_feats, text_feats,
                 extra_feats=None):
        """`extra_feats` (not in the reference): ordered {name: ndarray [n_items, d_name]} of further item modalities
        (BASELINE configs[1] names an acoustic one). Each gets `<name>_trans` / `<name>_embedding` modules created
        AFTER all reference modules (so a seed still reproduces the reference's initial weights) and is handled by
        `forward(..., extra_graphs=...)`. Parity of anything beyond image/text is UNPINNED: the reference loads
        image_feat.npy and text_feat.npy only (main.py:54-55)."""
        super().__init__()
        self.n_users = n_users
        self.n_items = n_items
        self.embedding_dim = embedding_dim
        self.n_ui_layers = len(weight_size)
        self.weight_size = [embedding_dim] + list(weight_size)
        d = args.embed_size
        # creation order == the reference's, so torch.manual_seed(s) reproduces its initial weights
        self.image_trans = nn.Linear(image_feats.shape[1], d)
        self.text_trans = nn.Linear(text_feats.shape[1], d)
        nn.init.xavier_uniform_(self.image_trans.weight)
        nn.init.xavier_uniform_(self.text_trans.weight)
        self.encoder = nn.ModuleDict({"image_encoder": self.image_trans, "text_encoder": self.text_trans})
        self.common_trans = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.common_trans.weight)
        self.align = nn.ModuleDict({"common_trans": self.common_trans})
        self.user_id_embedding = nn.Embedding(n_users, embedding_dim)
        self.item_id_embedding = nn.Embedding(n_items, embedding_dim)
        nn.init.xavier_uniform_(self.user_id_embedding.weight)
        nn.init.xavier_uniform_(self.item_id_embedding.weight)
        # raw modality features: plain attributes (not buffers) exactly like the reference
        self.image_feats = torch.as_tensor(image_feats).float()
        self.text_feats = torch.as_tensor(text_feats).float()
        self.image_embedding = nn.Embedding.from_pretrained(torch.Tensor(image_feats), freeze=False)
        self.text_embedding = nn.Embedding.from_pretrained(torch.Tensor(text_feats), freeze=False)
        self.batch_norm = nn.BatchNorm1d(d)
        self.tau = 0.5
        init = nn.init.xavier_uniform_
        self.weight_dict = nn.ParameterDict({
            "w_q": nn.Parameter(init(torch.empty([d, d]))),
            "w_k": nn.Parameter(init(torch.empty([d, d]))),
            "w_v": nn.Parameter(init(torch.empty([d, d]))),
            "w_self_attention_item": nn.Parameter(init(torch.empty([d, d]))),
            "w_self_attention_user": nn.Parameter(init(torch.empty([d, d]))),
            "w_self_attention_cat": nn.Parameter(init(torch.empty([args.head_num * d, d]))),
        })
        # w_q: the reference's attention computes Q with it (Models.py:139-169), but K is a reshuffled Q and the softmax
        # weights cancel, so its gradient is exactly zero on the CPU (~1e-9 noise on a GPU) - NOT None: torch.optim.AdamW
        # therefore applies its decoupled weight decay to it every step (w_q <- w_q (1 - lr * 0.01); measured on G12: the
        # trained w_q equals that to 1.2e-7), while w_k / w_v / ... receive no gradient at all and never move. The forward
        # here does not read w_q (see _modality_fusion); the flag makes FusedAdamW hand it the same all-zero gradient, so a
        # trained checkpoint's w_q matches the reference's.
        self.weight_dict["w_q"].unused_grad_is_zero = True
        self.embedding_dict = {"user": {}, "item": {}}
        self.extra_names = []
        self.extra_feats = {}
        for name, feats in (extra_feats or {}).items():
            if name in ("image", "text") or not name.isidentifier():
                raise ValueError("extra modality name %r" % (name,))
            lin = nn.Linear(feats.shape[1], d)
            nn.init.xavier_uniform_(lin.weight)
            setattr(self, name + "_trans", lin)
            self.encoder[name + "_encoder"] = lin
            setattr(self, name + "_embedding", nn.Embedding.from_pretrained(torch.Tensor(feats), freeze=False))
            self.extra_feats[name] = torch.as_tensor(feats).float()
            self.extra_names.append(name)

    # the feature matrices follow the module across devices (the reference hard-codes .cuda())
    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        self.image_feats = fn(self.image_feats)
        self.text_feats = fn(self.text_feats)
        self.extra_feats = {k_: fn(v) for k_, v in self.extra_feats.items()}
        return self

    # ---- helpers with the reference's names --------------------------------------------------
    def mm(self, x, y):
        """Models.py:69-73: sparse (or dense, args.sparse == 0) matrix product."""
        if args.sparse:
            return ops.spmm(_plan_of(x), y)
        return torch.mm(x, y)

    def sim(self, z1, z2):
        return torch.mm(ops.l2norm_rows(z1), ops.l2norm_rows(z2).t())

    def batched_contrastive_loss(self, z1, z2, batch_size=4096):
        """Models.py:79-98 (tau = self.tau, no +1e-8): kept for API parity; the trainer's variant
        (main.py:218-249) is the one on the hot path."""
        n = z1.size(0)
        f = lambda x: torch.exp(x / self.tau)     # noqa: E731
        losses = []
        for a in range(0, n, batch_size):
            refl = f(self.sim(z1[a:a + batch_size], z1))
            betw = f(self.sim(z1[a:a + batch_size], z2))
            k = torch.arange(refl.shape[0], device=z1.device)
            losses.append(-torch.log(betw[k, a + k] / (refl.sum(1) + betw.sum(1) - refl[k, a + k])))
        return torch.cat(losses).mean()

    def _modality_fusion(self, emb_a, emb_b):
        """mean over the two modality views of the reference's multi_head_self_attention
        (Models.py:139-169, 192-195). Because K is a reshuffled Q and V is aligned with the query
        axis, the softmax weights sum to 1 and each head returns V itself, i.e.
            Z_b = V_b @ (sum of the head_num row blocks of w_self_attention_cat)
        (the oracle keeps the literal 5-D form and tests bound the gap, <= 4e-6).
        w_q only ever receives an all-zero (GPU: ~1e-9 numerical-noise) gradient in the reference and w_k none; here w_q
        is handed the zero gradient by the optimiser (weight decay only, see __init__) and w_k none."""
        d = args.embed_size
        wcat = self.weight_dict["w_self_attention_cat"]
        fold = wcat.view(args.head_num, d, d).sum(0)               # [d, d]
        w = (0.5 * fold).t().contiguous()                          # linear() takes [out, in]
        return ops.linear(emb_a + emb_b, w)

    def feat_sumsq(self, g_item_image, g_item_text, g_user_image, g_user_text):
        """|a|^2+|b|^2+|c|^2+|d|^2 for the feature regulariser. When the arguments are exactly the
        four modal feature tensors of the latest forward, the value fused into that forward is
        returned (same autograd graph); anything else is reduced explicitly."""
        cached = getattr(self, "_feat_sumsq", None)
        args4 = (g_item_image, g_item_text, g_user_image, g_user_text)
        if cached is not None and all(a is b for a, b in zip(cached[1], args4)):
            return cached[0]
        return ops.sumsq(g_item_image) + ops.sumsq(g_item_text) + ops.sumsq(g_user_image) + ops.sumsq(g_user_text)

    def _zeros(self, rows, like):
        """Cached all-zero [rows, d] tensor (the value of A.E for an empty graph A)."""
        key = (rows, like.shape[1], like.device)
        z = self._zero_cache.get(key) if hasattr(self, "_zero_cache") else None
        if z is None:
            if not hasattr(self, "_zero_cache"):
                self._zero_cache = {}
            z = torch.zeros((rows, like.shape[1]), dtype=torch.float32, device=like.device)
            self._zero_cache[key] = z
        return z

    # ---- M-modality forward (extra modalities present) ------------------------------------------
    def _forward_multi(self, ui, iu, modal_plans, keep_masks):
        """The forward for a LIST of modalities [image, text, *extra] out of differentiable HIP ops (ops.linear,
        ops.spmm, ops.l2norm_rows): the same computation as the fused two-modality path, one more projection /
        feature chain / id view / `+ r * normalize(.)` term per extra modality (oracle: forward_multi)."""
        names = ["image", "text"] + self.extra_names
        feats = [self.image_feats, self.text_feats] + [self.extra_feats[n] for n in self.extra_names]
        p = float(args.drop_rate)
        scale, masks = 1.0, [None] * len(names)
        if self.training and p > 0.0:
            scale = 1.0 / (1.0 - p)
            masks = keep_masks if keep_masks is not None else list(ops.dropout_masks(
                len(names), self.n_items, args.embed_size, p, self.image_trans.weight.device))
        E_u, E_i = self.user_id_embedding.weight, self.item_id_embedding.weight
        user_f, item_f, user_id, item_id = [], [], [], []
        for k, nm in enumerate(names):
            lin = getattr(self, nm + "_trans")
            x = ops.linear(feats[k], lin.weight, lin.bias, masks[k], scale)
            uf = ops.spmm(ui, x)
            user_f.append(uf)
            item_f.append(ops.spmm(iu, uf))
            m_ui, m_iu = modal_plans[k]
            user_id.append(ops.spmm(m_ui, E_i))
            item_id.append(ops.spmm(m_iu, E_u))
        d = args.embed_size
        fold = self.weight_dict["w_self_attention_cat"].view(args.head_num, d, d).sum(0)
        w = ((1.0 / len(names)) * fold).t().contiguous()          # mean over the M views of V_b . fold
        u = ops.l2norm_rows(ops.linear(sum(user_id[1:], user_id[0]), w), E_u, args.id_cat_rate)
        i = ops.l2norm_rows(ops.linear(sum(item_id[1:], item_id[0]), w), E_i, args.id_cat_rate)
        us, its = u, i
        for l in range(self.n_ui_layers):
            epi = ops.EPI_SOFTMAX if l == self.n_ui_layers - 1 else ops.EPI_NONE
            u = ops.spmm(ui, i, epi)
            i = ops.spmm(iu, u, epi)
            us, its = us + u, its + i
        inv = 1.0 / (self.n_ui_layers + 1)
        u_g, i_g = us * inv, its * inv
        for uf, itf in zip(user_f, item_f):
            u_g = ops.l2norm_rows(uf, u_g, args.model_cat_rate)
            i_g = ops.l2norm_rows(itf, i_g, args.model_cat_rate)
        self._feat_sumsq = None
        for k, nm in enumerate(names):
            self.embedding_dict["user"][nm], self.embedding_dict["item"][nm] = user_id[k], item_id[k]
        out = [u_g, i_g, item_f[0], item_f[1], user_f[0], user_f[1], u_g, i_g, user_id[0], user_id[1], item_id[0],
               item_id[1]]
        for k in range(2, len(names)):
            out += [item_f[k], user_f[k], user_id[k], item_id[k]]
        return tuple(out)

    # ---- forward -----------------------------------------------------------------------------
    def hot_ctx(self):
        """The hotnode.HotCtx (side streams + hand-offs) of plain model(...) calls; a step object passes its own."""
        dev = self.user_id_embedding.weight.device
        c = getattr(self, "_hot_ctx", None)
        if c is None or c.device != dev:
            from .hotnode import HotCtx
            c = self._hot_ctx = HotCtx(dev)
        return c

    def forward(self, ui_graph, iu_graph, image_ui_graph, image_iu_graph, text_ui_graph, text_iu_graph,
                keep_masks=None, extra_graphs=None, hot=None):
        """Returns the 12-tuple of Models.py:220. `keep_masks` injects uint8 dropout keep-masks (parity runs): a pair
        (img, txt) of [n_items, d] tensors or one [2, n_items, d] tensor; by default the masks are drawn inside the
        projection's epilogue by the generator of ops.dropout_masks (seed with ops.seed_dropout / main.set_seed) in
        training mode. With extra modalities (ctor `extra_feats`) the tuple is followed by (item_feats, user_feats,
        user_id, item_id) of each extra modality; `extra_graphs` = {name: (ui_graph, iu_graph)} are their modal graphs
        (default: the interaction graphs, like the reference's initial image / text graphs, main.py:69-72).
        `hot`: the hotnode.HotCtx of a step object that owns the whole step (default: this model's own)."""
        from . import hotnode
        ui, iu = _plan_of(ui_graph), _plan_of(iu_graph)
        d = args.embed_size
        packed = hotnode.packed_supported([self.image_feats.shape[1], self.text_feats.shape[1]], self.n_items, d)
        if self.extra_names or not packed:
            # the modality LIST out of differentiable HIP ops (a third modality, or shapes the packed node does not take)
            modal = [(_plan_of(image_ui_graph).twin(3), _plan_of(image_iu_graph).twin(3)),
                     (_plan_of(text_ui_graph).twin(3), _plan_of(text_iu_graph).twin(3))]
            for nm in self.extra_names:
                g = (extra_graphs or {}).get(nm, (ui_graph, iu_graph))
                modal.append((_plan_of(g[0]).twin(3), _plan_of(g[1]).twin(3)))
            return self._forward_multi(ui, iu, modal, keep_masks)
        hot = hot if hot is not None else self.hot_ctx()
        img_ui, img_iu = _plan_of(image_ui_graph), _plan_of(image_iu_graph)
        txt_ui, txt_iu = _plan_of(text_ui_graph), _plan_of(text_iu_graph)
        # The modal-id SpMMs (and their autograd backward) run on the caller's stream while the hot node's chains may
        # still be running on its side streams (deferred join, hotpath.HotPathStep). A caller may pass the SAME plan as
        # ui_graph and image_ui_graph (Trainer's initial state, main.py:69-72): give the modal launches their own
        # partial-sum workspace + arrival counters (lanes 0 and 2 belong to the hot node).
        img_ui, img_iu, txt_ui, txt_iu = img_ui.twin(3), img_iu.twin(3), txt_ui.twin(3), txt_iu.twin(3)
        p = float(args.drop_rate)
        keep, scale, p_draw = None, 1.0, 0.0
        if self.training and p > 0.0:
            scale = 1.0 / (1.0 - p)
            if keep_masks is None:
                p_draw = p
            elif torch.is_tensor(keep_masks):
                keep = keep_masks.contiguous()
            else:
                keep = torch.stack(tuple(keep_masks)).contiguous()
        E_u, E_i = self.user_id_embedding.weight, self.item_id_embedding.weight
        # the reference repeats this block args.layers times without feeding anything back
        # (Models.py:176-186): the result is that of one pass.
        assert args.layers >= 1
        wcat = self.weight_dict["w_self_attention_cat"]
        # Modal id views. An EMPTY modal graph (the reference's state from the third batch on,
        # SURVEY 8a-3) makes A.E == 0, the fused view 0 and normalize(0) == 0, so u == E_u exactly:
        # that branch is skipped; w_self_attention_cat still receives its exactly-zero gradient.
        users_empty = img_ui.nnz == 0 and txt_ui.nnz == 0
        items_empty = img_iu.nnz == 0 and txt_iu.nnz == 0
        if users_empty:
            image_user_id = text_user_id = self._zeros(self.n_users, E_u)
            u = ops.zero_grad_anchor(E_u, wcat, hot)
        else:
            image_user_id = ops.spmm(img_ui, E_i)
            text_user_id = ops.spmm(txt_ui, E_i)
            u = ops.l2norm_rows(self._modality_fusion(image_user_id, text_user_id), E_u, args.id_cat_rate)
        if items_empty:
            image_item_id = text_item_id = self._zeros(self.n_items, E_i)
            i = E_i if users_empty else ops.zero_grad_anchor(E_i, wcat, hot)    # one anchor is enough
        else:
            image_item_id = ops.spmm(img_iu, E_u)
            text_item_id = ops.spmm(txt_iu, E_u)
            i = ops.l2norm_rows(self._modality_fusion(image_item_id, text_item_id), E_i, args.id_cat_rate)
        self.embedding_dict["user"]["image"] = image_user_id
        self.embedding_dict["user"]["text"] = text_user_id
        self.embedding_dict["item"]["image"] = image_item_id
        self.embedding_dict["item"]["text"] = text_item_id

        # projection of both modalities (one grouped launch), packed modal SpMM chain, G-layer propagation, layer mean
        # and "+ rate * normalize(modal feats)" as one fused node on two forked streams; its by-product `ss` is the
        # feature regulariser's sum of squares (main.py:252-257)
        lins = (self.image_trans, self.text_trans)
        u_g, i_g, ss, MI, MU = hotnode.hot_node(
            hot, (self.image_feats, self.text_feats), [l.weight for l in lins], [l.bias for l in lins], keep, p_draw, scale,
            u, i, ui, iu, self.n_ui_layers, args.model_cat_rate)
        # the reference's four modal feature outputs are column slices of the packed tables
        image_item_feats, text_item_feats = MI[:, :d], MI[:, d:2 * d]
        image_user_feats, text_user_feats = MU[:, :d], MU[:, d:2 * d]
        self._feat_sumsq = (ss, (image_item_feats, text_item_feats, image_user_feats, text_user_feats))
        return (u_g, i_g, image_item_feats, text_item_feats, image_user_feats, text_user_feats, u_g, i_g,
                image_user_id, text_user_id, image_item_id, text_item_id)


class Discriminator(nn.Module):
    """WGAN critic over n_items-wide rows (Models.py:224-245). Dense MLP: stays on stock
    PyTorch-ROCm ops (out of the hot-path scope). `nn.LeakyReLU(True)` in the reference means
    negative_slope == 1.0, i.e. the identity; reproduced as such."""

    def __init__(self, dim):
        super().__init__()
        self.net = nn.Sequential(
            nn.Linear(dim, int(dim / 4)), nn.LeakyReLU(1.0), nn.BatchNorm1d(int(dim / 4)), nn.Dropout(args.G_drop1),
            nn.Linear(int(dim / 4), int(dim / 8)), nn.LeakyReLU(1.0), nn.BatchNorm1d(int(dim / 8)),
            nn.Dropout(args.G_drop2),
            nn.Linear(int(dim / 8), 1), nn.Sigmoid())

    def forward(self, x):
        return (100 * self.net(x.float())).view(-1)
