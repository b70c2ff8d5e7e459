"""CPU ORACLE for the MMSSL hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch torch-CPU / numpy restatement of the reference algorithm for the path
named by BASELINE.json `north_star` (HKUDS/MMSSL). It exists so that the HIP kernels can
be checked on the GPU box, where /root/reference does not exist. Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the product
package `mmssl_amd` never does (tests/test_boundary.py greps for that).

Parity status: PINNED. Every function below is checked against golden vectors produced
by running the upstream Python reference itself in the build container
(oracle/gen_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).
The reference has no tests or golden vectors of its own.

Reference line numbers are relative to /root/reference/MMSSL/.
All arithmetic is fp32 like the reference; indices are int64.
"""
import math
import random as _pyrandom

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# graph preparation (host)                                       main.py:89-112, 513-520
# --------------------------------------------------------------------------------------
def csr_norm(csr_mat, mean_flag=False):
    """diag((rowsum+1e-8)^-1/2) * A  [* diag((colsum+1e-8)^-1/2) when mean_flag is False].

    main.py:89-103. With mean_flag=True (the only mode the trainer uses, main.py:66-67)
    an edge (r, c) is scaled by 1/sqrt(deg(r)); empty rows get 1e4 * 0 = nothing.
    """
    rs = np.asarray(csr_mat.sum(1)).flatten()
    rs = np.power(rs + 1e-8, -0.5)
    rs[np.isinf(rs)] = 0.0
    left = sp.diags(rs)
    if mean_flag:
        return left * csr_mat
    cs = np.asarray(csr_mat.sum(0)).flatten()
    cs = np.power(cs + 1e-8, -0.5)
    cs[np.isinf(cs)] = 0.0
    return left * csr_mat * sp.diags(cs)


def to_torch_sparse(mat):
    """scipy sparse -> torch sparse COO fp32 (uncoalesced, rows sorted). main.py:105-112."""
    coo = mat.tocoo()
    idx = torch.from_numpy(np.vstack((coo.row, coo.col)).astype(np.int64))
    val = torch.from_numpy(coo.data.astype(np.float32))
    return torch.sparse_coo_tensor(idx, val, torch.Size(coo.shape)).to(torch.float32)


def graph_pair(ui_raw):
    """(A_ui, A_iu) as the trainer builds them: each row-normalised by its OWN degrees
    (A_iu is NOT A_ui^T). main.py:65-67."""
    return to_torch_sparse(csr_norm(ui_raw, True)), to_torch_sparse(csr_norm(ui_raw.T, True))


def spmm(A, X):
    """Y = A @ X, A sparse [R, C], X dense [C, d]. Models.py:69-73, 203-208."""
    return torch.sparse.mm(A, X)


# --------------------------------------------------------------------------------------
# model forward                                                        Models.py:139-220
# --------------------------------------------------------------------------------------
class Cfg:
    """The live flags on the hot path and their defaults (utility/parser.py)."""

    def __init__(self, embed_size=64, head_num=4, layers=1, n_ui_layers=2, drop_rate=0.2,
                 model_cat_rate=0.55, id_cat_rate=0.36, tau=0.5, cl_rate=0.03,
                 feat_reg_decay=1e-5, decay=1e-5, batch_size=1024, G_rate=1e-4):
        self.__dict__.update(locals())
        del self.__dict__["self"]


def modality_attention(P, emb_a, emb_b, cfg):
    """The reference's "multi-head self attention" over the two modality views, restated
    faithfully (Models.py:139-169) — including its quirks:
      * K is a re-interpretation of the permuted Q buffer, w_k is unused   (:150)
      * V's modality axis lines up with the *query* axis, so the softmax
        weights are summed to 1 and every head returns V itself          (:154-162)
    Returns Z [2, N, d]. The product folds this to V @ (sum of the 4 row blocks of
    w_self_attention_cat); tests bound the difference.
    """
    H, d = cfg.head_num, cfg.embed_size
    dh = d // H
    V = torch.stack([emb_a, emb_b], 0)                       # [beh, N, d]
    beh, N = V.shape[0], V.shape[1]
    Q = V @ P["weight_dict.w_q"]
    Qp = Q.reshape(beh, N, H, dh).permute(2, 0, 1, 3)        # [H, beh, N, dh]
    Kp = Qp.contiguous().view(beh, N, H, dh).permute(2, 0, 1, 3)
    att = (Qp.unsqueeze(2) * Kp.unsqueeze(1)).sum(-1) / math.sqrt(float(d) / H)  # [H,bq,bk,N]
    att = torch.softmax(att, dim=2).unsqueeze(-1)            # over bk
    Z = (att * V.unsqueeze(1)).sum(2)                        # [H, bq, N, d]; V indexed by bq
    Z = torch.cat([Z[h] for h in range(H)], -1)              # [beh, N, H*d]
    return Z @ P["weight_dict.w_self_attention_cat"]


def forward(P, image_feats, text_feats, graphs, cfg, training=False, keep_masks=None):
    """MMSSL.forward (Models.py:171-220). `graphs` = (ui, iu, img_ui, img_iu, txt_ui, txt_iu)
    torch sparse tensors. `keep_masks` = optional (mask_img, mask_txt) 0/1 tensors [I, d]:
    dropout keeps where 1 and scales by 1/(1-p) (nn.Dropout, Models.py:54,173-174).
    Returns the 12-tuple of Models.py:220 (items 0/6 and 1/7 are the same tensors)."""
    ui, iu, img_ui, img_iu, txt_ui, txt_iu = graphs
    x_img = F.linear(image_feats, P["image_trans.weight"], P["image_trans.bias"])
    x_txt = F.linear(text_feats, P["text_trans.weight"], P["text_trans.bias"])
    if training and cfg.drop_rate > 0:
        assert keep_masks is not None, "oracle needs injected dropout masks in training mode"
        s = 1.0 / (1.0 - cfg.drop_rate)
        x_img = x_img * keep_masks[0] * s
        x_txt = x_txt * keep_masks[1] * s
    E_u, E_i = P["user_id_embedding.weight"], P["item_id_embedding.weight"]
    assert cfg.layers >= 1
    for _ in range(cfg.layers):          # body never feeds back: idempotent (Models.py:176-186)
        img_user = spmm(ui, x_img)
        img_item = spmm(iu, img_user)
        img_user_id = spmm(img_ui, E_i)
        img_item_id = spmm(img_iu, E_u)
        txt_user = spmm(ui, x_txt)
        txt_item = spmm(iu, txt_user)
        txt_user_id = spmm(txt_ui, E_i)
        txt_item_id = spmm(txt_iu, E_u)
    user_z = modality_attention(P, img_user_id, txt_user_id, cfg).mean(0)
    item_z = modality_attention(P, img_item_id, txt_item_id, cfg).mean(0)
    u = E_u + cfg.id_cat_rate * F.normalize(user_z, p=2, dim=1)
    i = E_i + cfg.id_cat_rate * F.normalize(item_z, p=2, dim=1)
    us, its = [u], [i]
    for l in range(cfg.n_ui_layers):
        u = spmm(ui, i)
        if l == cfg.n_ui_layers - 1:
            u = torch.softmax(u, dim=-1)       # over the feature dim, last layer only (:202-204)
        i = spmm(iu, u)                        # uses the already-updated (softmaxed) u
        if l == cfg.n_ui_layers - 1:
            i = torch.softmax(i, dim=-1)
        us.append(u)
        its.append(i)
    u = torch.stack(us).mean(0)
    i = torch.stack(its).mean(0)
    r = cfg.model_cat_rate
    u = u + r * F.normalize(img_user, p=2, dim=1) + r * F.normalize(txt_user, p=2, dim=1)
    i = i + r * F.normalize(img_item, p=2, dim=1) + r * F.normalize(txt_item, p=2, dim=1)
    return (u, i, img_item, txt_item, img_user, txt_user, u, i,
            img_user_id, txt_user_id, img_item_id, txt_item_id)


def modality_attention_multi(P, embs, cfg):
    """modality_attention over M >= 2 modality views (the reference's code is already written for a dict of
    views, Models.py:139-169; it only ever holds two). Same quirks, same operation order."""
    H, d = cfg.head_num, cfg.embed_size
    dh = d // H
    V = torch.stack(list(embs), 0)                           # [beh, N, d]
    beh, N = V.shape[0], V.shape[1]
    Q = V @ P["weight_dict.w_q"]
    Qp = Q.reshape(beh, N, H, dh).permute(2, 0, 1, 3)
    Kp = Qp.contiguous().view(beh, N, H, dh).permute(2, 0, 1, 3)
    att = (Qp.unsqueeze(2) * Kp.unsqueeze(1)).sum(-1) / math.sqrt(float(d) / H)
    att = torch.softmax(att, dim=2).unsqueeze(-1)
    Z = (att * V.unsqueeze(1)).sum(2)
    Z = torch.cat([Z[h] for h in range(H)], -1)
    return Z @ P["weight_dict.w_self_attention_cat"]


def forward_multi(P, feats, graphs, modal_graphs, cfg, names=("image", "text"), training=False, keep_masks=None):
    """MMSSL.forward generalised to a LIST of modalities (BASELINE configs[1] names V/A/T).
    With names == ("image", "text") this is `forward` operation for operation (tests pin that, bit for bit); a third
    modality follows the same pattern: its own projection `<name>_trans`, feature chain, id views from its own
    modal graph pair, one more view in the attention mean, one more `+ r * normalize(.)` term.
    PARITY STATUS of any modality beyond image/text: UNPINNED - the reference loads only image_feat.npy and
    text_feat.npy (main.py:54-55), so there is nothing to pin an acoustic branch to.
    feats: list of [I, d_m]; graphs = (ui, iu); modal_graphs: list of (m_ui, m_iu). Returns a dict."""
    ui, iu = graphs
    E_u, E_i = P["user_id_embedding.weight"], P["item_id_embedding.weight"]
    xs = []
    for k, (nm, Fm) in enumerate(zip(names, feats)):
        x = F.linear(Fm, P[nm + "_trans.weight"], P[nm + "_trans.bias"])
        if training and cfg.drop_rate > 0:
            assert keep_masks is not None
            x = x * keep_masks[k] * (1.0 / (1.0 - cfg.drop_rate))
        xs.append(x)
    user_f, item_f, user_id, item_id = [], [], [], []
    for x, (m_ui, m_iu) in zip(xs, modal_graphs):
        uf = spmm(ui, x)
        user_f.append(uf)
        item_f.append(spmm(iu, uf))
        user_id.append(spmm(m_ui, E_i))
        item_id.append(spmm(m_iu, E_u))
    user_z = modality_attention_multi(P, user_id, cfg).mean(0)
    item_z = modality_attention_multi(P, item_id, cfg).mean(0)
    u = E_u + cfg.id_cat_rate * F.normalize(user_z, p=2, dim=1)
    i = E_i + cfg.id_cat_rate * F.normalize(item_z, p=2, dim=1)
    us, its = [u], [i]
    for l in range(cfg.n_ui_layers):
        u = spmm(ui, i)
        if l == cfg.n_ui_layers - 1:
            u = torch.softmax(u, dim=-1)
        i = spmm(iu, u)
        if l == cfg.n_ui_layers - 1:
            i = torch.softmax(i, dim=-1)
        us.append(u)
        its.append(i)
    u = torch.stack(us).mean(0)
    i = torch.stack(its).mean(0)
    r = cfg.model_cat_rate
    for uf, itf in zip(user_f, item_f):
        u = u + r * F.normalize(uf, p=2, dim=1)
        i = i + r * F.normalize(itf, p=2, dim=1)
    return {"ua": u, "ia": i, "item_feats": item_f, "user_feats": user_f, "user_id": user_id, "item_id": item_id}


def generator_loss_multi(o, users, pos, neg, n_items, cfg):
    """main.py:368-371, 407-420 without the GAN term, over every modality of a forward_multi result."""
    mf, emb, _ = bpr(o["ua"][users], o["ia"][pos], o["ia"][neg], cfg.decay, cfg.batch_size)
    s = 0.0
    for t in o["item_feats"] + o["user_feats"]:
        s = s + 0.5 * (t ** 2).sum()
    feat = cfg.feat_reg_decay * (s / n_items)
    cl = 0.0
    for z in o["user_id"]:
        cl = cl + infonce(z[users], o["ua"][users], cfg.tau)
    return mf + emb + feat + cfg.cl_rate * cl


def gcn_propagate(ui, iu, u0, i0, n_layers):
    """Just the K4 chain (Models.py:199-214): the '3-layer GCN SpMM' the metric counts."""
    u, i = u0, i0
    us, its = [u], [i]
    for l in range(n_layers):
        u = spmm(ui, i)
        if l == n_layers - 1:
            u = torch.softmax(u, dim=-1)
        i = spmm(iu, u)
        if l == n_layers - 1:
            i = torch.softmax(i, dim=-1)
        us.append(u)
        its.append(i)
    return torch.stack(us).mean(0), torch.stack(its).mean(0)


# --------------------------------------------------------------------------------------
# losses                                                    main.py:211-257, 499-511
# --------------------------------------------------------------------------------------
def infonce(z1, z2, tau=0.5, block=1024):
    """Trainer.batched_contrastive_loss (main.py:218-249) with sim (:211-216), evaluated in
    the same block structure (row blocks x concatenated column blocks of `block`)."""
    n = z1.shape[0]
    n1, n2 = F.normalize(z1), F.normalize(z2)
    losses = []
    for a in range(0, n, block):
        rows = slice(a, min(a + block, n))
        refl = torch.cat([torch.exp(n1[rows] @ n1[b:b + block].t() / tau) for b in range(0, n, block)], -1)
        betw = torch.cat([torch.exp(n1[rows] @ n2[b:b + block].t() / tau) for b in range(0, n, block)], -1)
        idx = torch.arange(rows.start, rows.stop)
        loc = torch.arange(rows.stop - rows.start)
        pos = betw[loc, idx]
        denom = refl.sum(1) + betw.sum(1) - refl[loc, idx]
        losses.append(-torch.log(pos / denom + 1e-8))        # +1e-8 is INSIDE the log (:244)
    return torch.cat(losses).mean()


def bpr(u, p, n, decay, batch_size):
    """Trainer.bpr_loss (main.py:499-511): (mf_loss, emb_loss, reg_loss=0.0)."""
    diff = (u * p).sum(1) - (u * n).sum(1)
    mf = -F.logsigmoid(diff).mean()
    reg = 0.5 * (u ** 2).sum() + 0.5 * (p ** 2).sum() + 0.5 * (n ** 2).sum()
    return mf, decay * (reg / batch_size), 0.0


def feat_reg(item_img, item_txt, user_img, user_txt, n_items, feat_reg_decay):
    """Trainer.feat_reg_loss_calculation (main.py:252-257)."""
    s = 0.5 * (item_img ** 2).sum() + 0.5 * (item_txt ** 2).sum() \
        + 0.5 * (user_img ** 2).sum() + 0.5 * (user_txt ** 2).sum()
    return feat_reg_decay * (s / n_items)


def u_sim(users, user_final, item_final, ui_raw, batch_size):
    """Trainer.u_sim_calculation (main.py:283-298): masked, row-L2-normalised user x item
    scores (adjacent to the hot path; needed for the G-step assembly check)."""
    tu = user_final[users]
    seen = torch.from_numpy(np.asarray(ui_raw[np.asarray(users)].todense(), dtype=np.float32))
    n_items = item_final.shape[0]
    chunks = []
    for a in range(0, n_items, batch_size):
        s = tu @ item_final[a:a + batch_size].t()
        chunks.append(s * (1 - seen[:, a:a + batch_size]))
    return F.normalize(torch.cat(chunks, -1), p=2, dim=1)


def discriminator_eval(Dsd, x):
    """Discriminator.forward in eval() mode (Models.py:224-245). LeakyReLU(True) has
    negative_slope = 1.0, i.e. identity; BatchNorm uses running stats; Dropout is off."""
    def bn(h, k):
        return F.batch_norm(h, Dsd["net.%d.running_mean" % k], Dsd["net.%d.running_var" % k],
                            Dsd["net.%d.weight" % k], Dsd["net.%d.bias" % k], False, 0.0, 1e-5)
    h = F.linear(x.float(), Dsd["net.0.weight"], Dsd["net.0.bias"])
    h = bn(h, 2)
    h = F.linear(h, Dsd["net.4.weight"], Dsd["net.4.bias"])
    h = bn(h, 6)
    h = torch.sigmoid(F.linear(h, Dsd["net.8.weight"], Dsd["net.8.bias"]))
    return (100 * h).view(-1)


def g_step_losses(P, Dsd, image_feats, text_feats, graphs, ui_raw, users, pos, neg, cfg):
    """The generator-step loss assembly (main.py:363-420) with dropout off and D frozen in
    eval mode. Returns dict of component losses and the total `batch_loss`."""
    o = forward(P, image_feats, text_feats, graphs, cfg, training=False)
    ua, ia, img_item, txt_item, img_user, txt_user, uemb, _, img_uid, txt_uid, _, _ = o
    mf, emb, reg = bpr(ua[users], ia[pos], ia[neg], cfg.decay, cfg.batch_size)
    s_img = u_sim(users, img_user, img_item, ui_raw, cfg.batch_size)
    s_txt = u_sim(users, txt_user, txt_item, ui_raw, cfg.batch_size)
    fr = feat_reg(img_item, txt_item, img_user, txt_user, ia.shape[0], cfg.feat_reg_decay)
    cl1 = infonce(img_uid[users], uemb[users], cfg.tau)
    cl2 = infonce(txt_uid[users], uemb[users], cfg.tau)
    g_lossf = -discriminator_eval(Dsd, torch.cat((s_img, s_txt), 0)).mean()
    total = mf + emb + reg + fr + cfg.cl_rate * (cl1 + cl2) + cfg.G_rate * g_lossf
    return dict(mf=mf, emb=emb, feat=fr, cl1=cl1, cl2=cl2, G_lossf=g_lossf, batch_loss=total)


# --------------------------------------------------------------------------------------
# BPR triple sampler (host, bit-exact)                     utility/load_data.py:153-191
# --------------------------------------------------------------------------------------
def set_seed(seed):
    """main.py:522-526 (the torch.cuda seeding is a no-op on CPU)."""
    np.random.seed(seed)
    _pyrandom.seed(seed)
    torch.manual_seed(seed)


def sample_bpr(exist_users, train_items, n_items, n_users, batch_size):
    """One Data.sample() call. Consumes the GLOBAL `random` and `np.random` streams in
    exactly the reference's order: users first (random.sample, or batch_size x
    random.choice when batch_size > n_users), then per user, interleaved, one positive
    (np.random.randint over the user's list, retried on duplicates) and one negative
    (np.random.randint over all items, rejected while in the user's train list)."""
    if batch_size <= n_users:
        users = _pyrandom.sample(exist_users, batch_size)
    else:
        users = [_pyrandom.choice(exist_users) for _ in range(batch_size)]
    pos, neg = [], []
    for u in users:
        mine = train_items[u]
        pos.append(mine[np.random.randint(low=0, high=len(mine), size=1)[0]])
        while True:
            cand = np.random.randint(low=0, high=n_items, size=1)[0]
            if cand not in mine:
                neg.append(cand)
                break
    return users, pos, neg


# --------------------------------------------------------------------------------------
# evaluation (Recall@K parity)          utility/batch_test.py:21-36,83-169, metrics.py
# --------------------------------------------------------------------------------------
def rank_hits(scores, train_items_u, pos_items_u, k_max):
    """Top-k_max hit list for one user: candidates = all items not in the user's train
    list, ascending id; ranked by score descending, ties keep ascending id
    (heapq.nlargest == stable sort; batch_test.py:21-36, 98-104)."""
    n_items = scores.shape[0]
    allowed = np.ones(n_items, bool)
    allowed[np.asarray(train_items_u, dtype=np.int64)] = False
    cand = np.nonzero(allowed)[0]
    order = np.argsort(-scores[cand], kind="stable")[:k_max]
    top = cand[order]
    posset = set(int(x) for x in pos_items_u)
    return [1 if int(t) in posset else 0 for t in top]


def metrics_at(r, k, n_pos):
    """precision/recall/ndcg/hit @k from a hit list r (metrics.py:8-99). NDCG's ideal DCG
    is computed from the *sorted hit list itself* (metrics.py:67-71), not from n_pos."""
    rk = np.asarray(r, dtype=np.float64)[:k]
    prec = float(np.mean(rk))
    rec = float(np.sum(rk) / n_pos) if n_pos else 0.0
    disc = np.log2(np.arange(2, rk.size + 2))
    dcg = float(np.sum(rk / disc))
    ideal = np.asarray(sorted(r, reverse=True), dtype=np.float64)[:k]
    idcg = float(np.sum(ideal / np.log2(np.arange(2, ideal.size + 2))))
    ndcg = dcg / idcg if idcg else 0.0
    hit = 1.0 if np.sum(rk) > 0 else 0.0
    return prec, rec, ndcg, hit


def evaluate(ua, ia, users, train_items, pos_sets, Ks):
    """test_torch (batch_test.py:112-169): score = ua[users] @ ia.T in fp32, per-user
    metrics averaged over len(users)."""
    res = {k: np.zeros(len(Ks)) for k in ("precision", "recall", "ndcg", "hit_ratio")}
    n = len(users)
    rate = (ua[users] @ ia.t()).detach().cpu().numpy()
    for row, u in zip(rate, users):
        r = rank_hits(row, train_items.get(u, []), pos_sets[u], max(Ks))
        for j, k in enumerate(Ks):
            p, rc, nd, h = metrics_at(r, k, len(pos_sets[u]))
            res["precision"][j] += p / n
            res["recall"][j] += rc / n
            res["ndcg"][j] += nd / n
            res["hit_ratio"][j] += h / n
    return res
