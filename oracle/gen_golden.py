"""Generate tests/golden/*.npz by RUNNING THE UPSTREAM REFERENCE on CPU — TEST INFRASTRUCTURE.

Run here only (the container that has /root/reference):   python oracle/gen_golden.py
The committed outputs are DATA (inputs + the reference's outputs); no reference source,
bytecode or text is stored. Vector ids follow SURVEY.md section 8c (G1..G8).

Each fixture cites the reference call that produced it:
  G1  Trainer.csr_norm(mean_flag=True)                  main.py:89-103
  G2  MMSSL.forward, eval mode                          Models.py:171-220
  G3  MMSSL.forward train mode (drop_rate=0) + backward Models.py:171-220
  G4  Trainer.batched_contrastive_loss                  main.py:211-249
  G5  Trainer.bpr_loss / feat_reg_loss_calculation      main.py:499-511, 252-257
  G6  Data.sample()                                     utility/load_data.py:153-191
  G7  test_torch (Recall/NDCG/precision/hit @ Ks)       utility/batch_test.py:112-169
  G8  G-step loss assembly                              main.py:363-420
  G9  modal-graph maintenance inside Trainer.train()    main.py:378-405 (k = int(n_items*m_topk_rate) in {1, 2}, T in {1, 2})
  G16 the same loop + evaluation at the Amazon-Baby SHAPE (python oracle/gen_golden.py g16; see gen_g16)
  G12 K-batch trajectory of Trainer.train() + evaluation  main.py:308-429 -> utility/batch_test.py:112-169
      (python oracle/gen_golden.py g12: own process, discriminator dropout off so that every random draw of the loop
       is one of the two recorded tensors)
"""
import os
import shutil
import sys

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
import synth_data  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden")
TMP = "/tmp/mmssl_golden/"
U, I, E, DV, DT, D, B = 160, 96, 1000, 32, 48, 64, 48


def npy(t):
    return t.detach().cpu().numpy()


def cotangent(k, shape):
    """Deterministic analytic cotangent for output k (recomputed, not stored, by the tests)."""
    i = np.arange(shape[0], dtype=np.float64)[:, None]
    j = np.arange(shape[1], dtype=np.float64)[None, :]
    return np.sin(0.37 * i + 1.3 * j + 0.71 * k).astype(np.float32)


def coo_triplets(csr):
    c = csr.tocoo()
    return c.row.astype(np.int64), c.col.astype(np.int64), c.data.astype(np.float32)


def params_of(model):
    sd = model.state_dict()
    keep = ["image_trans.weight", "image_trans.bias", "text_trans.weight", "text_trans.bias",
            "user_id_embedding.weight", "item_id_embedding.weight",
            "weight_dict.w_q", "weight_dict.w_k", "weight_dict.w_v",
            "weight_dict.w_self_attention_cat"]
    return {"p." + k: npy(sd[k]) for k in keep}


OUT_NAMES = ["ua", "ia", "image_item", "text_item", "image_user", "text_user",
             "ua2", "ia2", "image_user_id", "text_user_id", "image_item_id", "text_item_id"]


def main():
    if os.path.isdir(TMP):
        shutil.rmtree(TMP)
    os.makedirs(OUT, exist_ok=True)
    synth_data.write_dataset(TMP, "tiny", U, I, E, DV, DT, seed=1)
    # copy the dataset itself as a fixture (json + npy + triplets): tests rebuild it on disk
    ref = ref_shim.load(TMP, "tiny", ["--batch_size", str(B), "--drop_rate", "0.0"])
    Models = sys.modules["Models"]
    args = ref.args
    margs = Models.args
    dg = ref.data_generator

    # ---------------- G6: sampler, must come first (fresh RNG state) ----------------
    ref.set_seed(2022)
    batches = [dg.sample() for _ in range(3)]
    np.savez(os.path.join(OUT, "g6_sample.npz"),
             users=np.array([b[0] for b in batches], np.int64),
             pos=np.array([b[1] for b in batches], np.int64),
             neg=np.array([b[2] for b in batches], np.int64), seed=2022, batch_size=B)
    # B > n_users branch (random.choice path, load_data.py:156-157)
    dg.batch_size = U + 37
    ref.set_seed(7)
    big = dg.sample()
    dg.batch_size = B
    np.savez(os.path.join(OUT, "g6_sample_big.npz"), users=np.array(big[0], np.int64),
             pos=np.array(big[1], np.int64), neg=np.array(big[2], np.int64), seed=7,
             batch_size=U + 37)

    # dataset fixture (so tests can rebuild the on-disk dataset without synth code drift)
    import json, pickle
    ddir = os.path.join(TMP, "tiny")
    tm = pickle.load(open(os.path.join(ddir, "train_mat"), "rb"))
    r, c, v = coo_triplets(tm)

    def flat(js):
        ids, items, lens = [], [], []
        for k, lst in js.items():
            ids.append(int(k)); lens.append(len(lst)); items += lst
        return np.array(ids, np.int64), np.array(lens, np.int64), np.array(items, np.int64)

    ds = {}
    for nm in ("train", "val", "test"):
        a, b_, c_ = flat(json.load(open(os.path.join(ddir, nm + ".json"))))
        ds[nm + "_uid"], ds[nm + "_len"], ds[nm + "_items"] = a, b_, c_
    np.savez_compressed(os.path.join(OUT, "dataset_tiny.npz"), tm_row=r, tm_col=c, tm_val=v,
                        image_feat=np.load(os.path.join(ddir, "image_feat.npy")),
                        text_feat=np.load(os.path.join(ddir, "text_feat.npy")),
                        n_users=U, n_items=I, **ds)

    # ---------------- Trainer ----------------
    ref.set_seed(2022)
    tr = ref.Trainer(data_config={"n_users": dg.n_users, "n_items": dg.n_items})

    # ---------------- G1: csr_norm ----------------
    raw = tr.ui_graph_raw
    n_ui = tr.csr_norm(raw, mean_flag=True).tocsr()
    n_iu = tr.csr_norm(raw.T, mean_flag=True).tocsr()
    # custom matrix with empty rows and non-unit values
    rng = np.random.default_rng(5)
    cm = sp.random(17, 11, density=0.2, random_state=3, format="csr", dtype=np.float32)
    cm.data = (rng.random(cm.nnz).astype(np.float32) + 0.5)
    cm = cm.tolil(); cm[4, :] = 0; cm[9, :] = 0; cm = cm.tocsr(); cm.eliminate_zeros()
    n_cm = tr.csr_norm(cm, mean_flag=True).tocsr()
    n_cm_sym = tr.csr_norm(cm, mean_flag=False).tocsr()
    g1 = {}
    for nm, m in (("raw", raw.tocsr()), ("ui", n_ui), ("iu", n_iu), ("cm", cm), ("cm_norm", n_cm),
                  ("cm_sym", n_cm_sym)):
        rr, cc, vv = coo_triplets(m)
        g1[nm + "_row"], g1[nm + "_col"], g1[nm + "_val"] = rr, cc, vv
        g1[nm + "_shape"] = np.array(m.shape, np.int64)
    np.savez_compressed(os.path.join(OUT, "g1_csr_norm.npz"), **g1)

    # ---------------- modal graph variants ----------------
    def graph_pair(ui_csr):
        ui = tr.matrix_to_tensor(tr.csr_norm(ui_csr, mean_flag=True))
        iu = tr.matrix_to_tensor(tr.csr_norm(ui_csr.T, mean_flag=True))
        return ui, iu

    rngm = np.random.default_rng(11)
    us = rngm.choice(U, size=B, replace=False)
    sparse_img = sp.csr_matrix((np.ones(B, np.float32), (us, rngm.integers(0, I, B))), shape=(U, I))
    sparse_txt = sp.csr_matrix((np.ones(B, np.float32), (us, rngm.integers(0, I, B))), shape=(U, I))
    empty = sp.csr_matrix((np.zeros(0, np.float32), (np.zeros(0, int), np.zeros(0, int))), shape=(U, I))
    variants = {
        "full": (raw, raw),
        "sparse": (sparse_img, sparse_txt),
        "empty": (empty, empty),
    }

    def run_forward(model, modal, train_mode):
        img_ui, img_iu = graph_pair(variants[modal][0])
        txt_ui, txt_iu = graph_pair(variants[modal][1])
        model.train(train_mode)
        return model(tr.ui_graph, tr.iu_graph, img_ui, img_iu, txt_ui, txt_iu)

    def modal_triplets(modal):
        out = {}
        for nm, m in (("img", variants[modal][0]), ("txt", variants[modal][1])):
            rr, cc, vv = coo_triplets(m.tocsr())
            out["modal_%s_row" % nm], out["modal_%s_col" % nm], out["modal_%s_val" % nm] = rr, cc, vv
        return out

    # ---------------- G2 / G3 ----------------
    for tag, wsize, layers in (("g2_l1", [64, 64], 1), ("g3_l2", [64, 64, 64], 2)):
        margs.layers = layers
        torch.manual_seed(100 + layers)
        model = ref.MMSSL(U, I, D, list(wsize), [0.1] * len(wsize), tr.image_feats, tr.text_feats)
        P = params_of(model)
        for modal in ("full", "sparse", "empty"):
            with torch.no_grad():
                outs = run_forward(model, modal, train_mode=False)
            np.savez_compressed(
                os.path.join(OUT, "g2_forward_%s_%s.npz" % (tag, modal)),
                weight_size=np.array(wsize), layers=layers,
                **P, **modal_triplets(modal), same_0_6=bool(outs[0] is outs[6]), same_1_7=bool(outs[1] is outs[7]),
                **{"o." + n: npy(o) for n, o in zip(OUT_NAMES, outs) if n not in ("ua2", "ia2")})
        # G3: train mode, drop_rate = 0 (args.drop_rate parsed as 0.0), backward of fixed scalar
        for modal in ("full", "sparse"):
            model.zero_grad()
            outs = run_forward(model, modal, train_mode=True)
            cots = [torch.from_numpy(cotangent(k, tuple(o.shape))) for k, o in enumerate(outs)]
            # outs[0] is outs[6] and outs[1] is outs[7] (Models.py:220): cotangents add up
            scalar = sum((o * c).sum() for o, c in zip(outs, cots))
            scalar.backward()
            grads = {}
            for k, p in model.named_parameters():
                if ("p." + k) in P and p.grad is not None:
                    grads["g." + k] = npy(p.grad)
            np.savez_compressed(
                os.path.join(OUT, "g3_backward_%s_%s.npz" % (tag, modal)),
                weight_size=np.array(wsize), layers=layers, scalar=float(scalar),
                **P, **modal_triplets(modal),
                **grads)
    margs.layers = 1

    # ---------------- G4: InfoNCE ----------------
    g4 = {}
    gen = torch.Generator().manual_seed(4)
    cases = {
        "n64": (torch.randn(64, D, generator=gen), torch.randn(64, D, generator=gen)),
        "n1100_d32": (torch.randn(1100, 32, generator=gen) * 0.3, torch.randn(1100, 32, generator=gen) * 2.0),
        "zero_z1": (torch.zeros(64, D), torch.randn(64, D, generator=gen)),
        "n130_d128": (torch.randn(130, 128, generator=gen), torch.randn(130, 128, generator=gen)),
    }
    z = cases["n64"][0].clone(); z[5] = 0; z[17] = 0
    cases["some_zero_rows"] = (z, cases["n64"][1].clone())
    for nm, (z1, z2) in cases.items():
        z1 = z1.clone().requires_grad_(True); z2 = z2.clone().requires_grad_(True)
        loss = tr.batched_contrastive_loss(z1, z2)
        loss.backward()
        g4[nm + ".z1"], g4[nm + ".z2"] = npy(z1), npy(z2)
        g4[nm + ".loss"] = np.float32(loss.item())
        g4[nm + ".gz1"], g4[nm + ".gz2"] = npy(z1.grad), npy(z2.grad)
    g4["tau"] = np.float32(args.tau)
    np.savez_compressed(os.path.join(OUT, "g4_infonce.npz"), **g4)

    # ---------------- G5: BPR + feat reg ----------------
    gen = torch.Generator().manual_seed(5)
    u = (torch.randn(B, D, generator=gen) * 0.2).requires_grad_(True)
    p = (torch.randn(B, D, generator=gen) * 0.2).requires_grad_(True)
    n = (torch.randn(B, D, generator=gen) * 0.2).requires_grad_(True)
    mf, emb, reg = tr.bpr_loss(u, p, n)
    (mf + emb).backward()
    a = (torch.randn(I, D, generator=gen)).requires_grad_(True)
    b = (torch.randn(I, D, generator=gen)).requires_grad_(True)
    c = (torch.randn(U, D, generator=gen)).requires_grad_(True)
    d = (torch.randn(U, D, generator=gen)).requires_grad_(True)
    fr = tr.feat_reg_loss_calculation(a, b, c, d)
    fr.backward()
    np.savez_compressed(os.path.join(OUT, "g5_bpr_featreg.npz"), u=npy(u), p=npy(p), n=npy(n),
                        mf=np.float32(mf.item()), emb=np.float32(emb.item()), reg=np.float32(reg),
                        gu=npy(u.grad), gp=npy(p.grad), gn=npy(n.grad), batch_size=B, decay=tr.decay,
                        fa=npy(a), fb=npy(b), fc=npy(c), fd=npy(d), feat_reg=np.float32(fr.item()),
                        gfa=npy(a.grad), gfc=npy(c.grad), feat_reg_decay=args.feat_reg_decay, n_items=I)

    # ---------------- G7: evaluation ----------------
    gen = torch.Generator().manual_seed(9)
    ua = torch.randn(U, D, generator=gen)
    ia = torch.randn(I, D, generator=gen)
    ia[7] = ia[3]  # force exact score ties (tie rule: lower item id first)
    ia[50] = ia[3]
    g7 = {"ua": npy(ua), "ia": npy(ia), "Ks": np.array(eval(args.Ks))}
    for is_val, nm in ((True, "val"), (False, "test")):
        users = list((dg.val_set if is_val else dg.test_set).keys())
        res = ref.test_torch(ua, ia, users, is_val)
        for k in ("precision", "recall", "ndcg", "hit_ratio"):
            g7["%s.%s" % (nm, k)] = np.asarray(res[k], np.float64)
        g7["%s.users" % nm] = np.array(users, np.int64)
    np.savez_compressed(os.path.join(OUT, "g7_eval.npz"), **g7)

    # ---------------- G8: G-step loss assembly (main.py:363-420), D frozen in eval ----------------
    torch.manual_seed(8)
    model = ref.MMSSL(U, I, D, [64, 64], [0.1, 0.1], tr.image_feats, tr.text_feats)
    tr.model = model
    tr.D.eval()
    model.train()
    ref.set_seed(2022)
    users, pos, neg = dg.sample()
    for modal in ("full", "empty"):
        img_ui, img_iu = graph_pair(variants[modal][0])
        txt_ui, txt_iu = graph_pair(variants[modal][1])
        model.zero_grad()
        (G_ua, G_ia, G_image_item, G_text_item, G_image_user, G_text_user, G_user_emb, _, G_image_user_id,
         G_text_user_id, _, _) = model(tr.ui_graph, tr.iu_graph, img_ui, img_iu, txt_ui, txt_iu)
        mf, emb, reg = tr.bpr_loss(G_ua[users], G_ia[pos], G_ia[neg])
        G_image_u_sim = tr.u_sim_calculation(users, G_image_user, G_image_item)
        G_text_u_sim = tr.u_sim_calculation(users, G_text_user, G_text_item)
        feat = tr.feat_reg_loss_calculation(G_image_item, G_text_item, G_image_user, G_text_user)
        cl1 = tr.batched_contrastive_loss(G_image_user_id[users], G_user_emb[users])
        cl2 = tr.batched_contrastive_loss(G_text_user_id[users], G_user_emb[users])
        G_lossf = -(tr.D(torch.cat((G_image_u_sim, G_text_u_sim), dim=0)).mean())
        batch_loss = mf + emb + reg + feat + args.cl_rate * (cl1 + cl2) + args.G_rate * G_lossf
        batch_loss.backward()
        grads = {"g." + k: npy(p_.grad) for k, p_ in model.named_parameters()
                 if ("p." + k) in params_of(model) and p_.grad is not None}
        np.savez_compressed(
            os.path.join(OUT, "g8_gstep_%s.npz" % modal), **params_of(model), **modal_triplets(modal),
            **{"D." + k: npy(v_) for k, v_ in tr.D.state_dict().items()},
            users=np.array(users, np.int64), pos=np.array(pos, np.int64), neg=np.array(neg, np.int64),
            mf=np.float32(mf.item()), emb=np.float32(emb.item()), feat=np.float32(feat.item()),
            cl1=np.float32(cl1.item()), cl2=np.float32(cl2.item()), G_lossf=np.float32(G_lossf.item()),
            batch_loss=np.float32(batch_loss.item()), cl_rate=args.cl_rate, G_rate=args.G_rate, **grads)

    gen_g9(ref, dg)
    for f in sorted(os.listdir(OUT)):
        print("%9d  %s" % (os.path.getsize(os.path.join(OUT, f)), f))


class _StopTraining(Exception):
    pass


def gen_g9(ref, dg, n_batches=5):
    """G9: run the reference's own Trainer.train() (main.py:320-430) for `n_batches` batches with k >= 1 and
    record, per batch, what its modal-graph maintenance consumed and produced: the batch users, the two
    [B, n_items] similarity matrices that feed torch.topk (4th and 5th u_sim_calculation call of a batch,
    main.py:372-373) and the four modal graphs the model receives at the START of every batch (i.e. the state
    left by the previous batch's rebuild-or-collect branch). Nothing of the loop is restated here: the loop is
    the reference's, observed through wrappers around model.forward / u_sim_calculation."""
    args = ref.args
    old = (args.m_topk_rate, args.T, args.epoch)
    for tag, rate, T in (("k1_T1", 0.011, 1), ("k2_T1", 0.021, 1), ("k2_T2", 0.021, 2)):
        args.m_topk_rate, args.T, args.epoch = rate, T, 1
        k = int(I * rate)
        ref.set_seed(2022)
        tr = ref.Trainer(data_config={"n_users": dg.n_users, "n_items": dg.n_items})
        rec = {"k": k, "T": T, "m_topk_rate": rate, "n_batches": n_batches}
        state = {"fwd_calls": 0, "sim_calls": 0}
        orig_fwd, orig_sim = tr.model.forward, tr.u_sim_calculation

        def fwd(*graphs, _o=orig_fwd):
            c = state["fwd_calls"]
            state["fwd_calls"] += 1
            if c % 2 == 0:                       # first forward of batch c // 2
                b = c // 2
                if b >= n_batches:
                    raise _StopTraining()
                for nm, g in zip(("img_ui", "img_iu", "txt_ui", "txt_iu"), graphs[2:6]):
                    g = g.coalesce()
                    idx = g.indices().numpy()
                    rec["b%d.%s_row" % (b, nm)] = idx[0].astype(np.int64)
                    rec["b%d.%s_col" % (b, nm)] = idx[1].astype(np.int64)
                    rec["b%d.%s_val" % (b, nm)] = g.values().numpy().astype(np.float32)
            return _o(*graphs)

        def sim(users, ue, ie, _o=orig_sim):
            c = state["sim_calls"]
            state["sim_calls"] += 1
            out = _o(users, ue, ie)
            b, j = divmod(c, 5)
            if j == 3:
                rec["b%d.users" % b] = np.array(users, np.int64)
                rec["b%d.img_sim" % b] = npy(out)
            elif j == 4:
                rec["b%d.txt_sim" % b] = npy(out)
            return out

        tr.model.forward = fwd
        tr.u_sim_calculation = sim
        try:
            tr.train()
        except _StopTraining:
            pass
        np.savez_compressed(os.path.join(OUT, "g9_modal_rebuild_%s.npz" % tag), **rec)
    args.m_topk_rate, args.T, args.epoch = old


def gen_g12(ref, dg, n_batches=8):
    """G12: the reference's own Trainer.train() for `n_batches` batches (k = 1, T = 1: batches 0-1 on the interaction
    graph, batch 2 on the B-edge top-1 graph, batch 3+ on empty modal graphs), then its own Trainer.test() on the
    validation and test users. Recorded through wrappers (nothing of the loop is restated): the initial model /
    discriminator parameters, every batch of Data.sample(), every random tensor the loop draws (the Gumbel uniforms of
    main.py:350 and the gradient-penalty alpha of main.py:147; model dropout and discriminator dropout are 0, so these
    are ALL draws), per-batch loss components, the final parameters, the eval-mode embeddings and the metric dict."""
    args = ref.args
    old = (args.m_topk_rate, args.T, args.epoch)
    args.m_topk_rate, args.T, args.epoch = 0.011, 1, 1
    ref.set_seed(2022)
    tr = ref.Trainer(data_config={"n_users": dg.n_users, "n_items": dg.n_items})
    skip = ("encoder.", "align.")
    rec = {"n_batches": n_batches, "k": int(I * args.m_topk_rate), "T": 1, "m_topk_rate": args.m_topk_rate,
           "lr": args.lr, "D_lr": args.D_lr, "G_rate": args.G_rate, "cl_rate": args.cl_rate, "gp_rate": args.gp_rate}
    for k, v in tr.model.state_dict().items():
        if not k.startswith(skip):
            rec["m0." + k] = npy(v).copy()          # a copy: the numpy view would follow the training in place
    for k, v in tr.D.state_dict().items():
        rec["D0." + k] = npy(v).copy()
    st = {"fwd": 0, "D": 0, "cl": 0, "uni": 0, "alpha": 0, "sample": 0, "on": False}
    o_fwd, o_D, o_bpr, o_cl = tr.model.forward, tr.D.forward, tr.bpr_loss, tr.batched_contrastive_loss
    o_feat, o_gp, o_sample = tr.feat_reg_loss_calculation, tr.gradient_penalty, dg.sample
    o_uniform, o_rand = torch.Tensor.uniform_, torch.rand

    def fwd(*graphs):
        c = st["fwd"]
        st["fwd"] += 1
        if st["on"] and c % 2 == 0 and c // 2 >= n_batches:
            raise _StopTraining()
        return o_fwd(*graphs)

    def D_fwd(x):
        out = o_D(x)
        if st["on"]:
            b, j = divmod(st["D"], 4)         # D(inputf), D(inputr), D(interpolates) inside the penalty, D(G_inputf)
            st["D"] += 1
            rec["b%d.D%d_mean" % (b, j)] = np.float32(out.detach().mean().item())
        return out

    def bpr(u, p, n):
        mf, emb, reg = o_bpr(u, p, n)
        b = st["fwd"] // 2 - 1
        rec["b%d.mf" % b], rec["b%d.emb" % b] = np.float32(mf.item()), np.float32(emb.item())
        return mf, emb, reg

    def cl(z1, z2):
        out = o_cl(z1, z2)
        b, j = divmod(st["cl"], 2)
        st["cl"] += 1
        rec["b%d.cl%d" % (b, j + 1)] = np.float32(out.item())
        return out

    def feat(a, b_, c, d):
        out = o_feat(a, b_, c, d)
        rec["b%d.feat" % (st["fwd"] // 2 - 1)] = np.float32(out.item())
        return out

    def gp(D, xr, xf):
        out = o_gp(D, xr, xf)
        rec["b%d.gp" % (st["fwd"] // 2)] = np.float32(out.item())
        return out

    def sample():
        out = o_sample()
        b = st["sample"]
        st["sample"] += 1
        if b < n_batches:
            rec["b%d.users" % b] = np.array(out[0], np.int64)
            rec["b%d.pos" % b] = np.array(out[1], np.int64)
            rec["b%d.neg" % b] = np.array(out[2], np.int64)
        return out

    def uniform_(self, *a, **k):
        out = o_uniform(self, *a, **k)
        if st["on"]:
            rec["b%d.gumbel_u" % st["uni"]] = npy(out).copy()
            st["uni"] += 1
        return out

    def rand(*a, **k):
        out = o_rand(*a, **k)
        if st["on"]:
            rec["b%d.gp_alpha" % st["alpha"]] = npy(out).copy()
            st["alpha"] += 1
        return out

    tr.model.forward, tr.D.forward, tr.bpr_loss, tr.batched_contrastive_loss = fwd, D_fwd, bpr, cl
    tr.feat_reg_loss_calculation, tr.gradient_penalty, dg.sample = feat, gp, sample
    torch.Tensor.uniform_, torch.rand = uniform_, rand
    st["on"] = True
    try:
        tr.train()
    except _StopTraining:
        pass
    finally:
        st["on"] = False
        torch.Tensor.uniform_, torch.rand = o_uniform, o_rand
        dg.sample = o_sample
        tr.model.forward, tr.D.forward = o_fwd, o_D
    assert st["uni"] == n_batches and st["alpha"] == n_batches and st["D"] == 4 * n_batches, st
    for b in range(n_batches):
        G_lossf = -float(rec["b%d.D3_mean" % b])
        rec["b%d.G_lossf" % b] = np.float32(G_lossf)
        rec["b%d.loss_D" % b] = np.float32(-float(rec["b%d.D1_mean" % b]) + float(rec["b%d.D0_mean" % b])
                                           + args.gp_rate * float(rec["b%d.gp" % b]))
        rec["b%d.batch_loss" % b] = np.float32(
            float(rec["b%d.mf" % b]) + float(rec["b%d.emb" % b]) + float(rec["b%d.feat" % b])
            + args.cl_rate * (float(rec["b%d.cl1" % b]) + float(rec["b%d.cl2" % b])) + args.G_rate * G_lossf)
    for k, v in tr.model.state_dict().items():
        if not k.startswith(skip) and not k.startswith(("image_embedding", "text_embedding")):
            rec["m1." + k] = npy(v)
    for k, v in tr.D.state_dict().items():
        rec["D1." + k] = npy(v)
    for nm, g in zip(("img_ui", "img_iu", "txt_ui", "txt_iu"),
                     (tr.image_ui_graph, tr.image_iu_graph, tr.text_ui_graph, tr.text_iu_graph)):
        rec["final.%s_nnz" % nm] = int(g._nnz())
    tr.model.eval()
    with torch.no_grad():
        outs = tr.model(tr.ui_graph, tr.iu_graph, tr.image_ui_graph, tr.image_iu_graph, tr.text_ui_graph, tr.text_iu_graph)
    rec["eval.ua"], rec["eval.ia"] = npy(outs[0]), npy(outs[1])
    for is_val, nm in ((True, "val"), (False, "test")):
        users = list((dg.val_set if is_val else dg.test_set).keys())
        res = tr.test(users, is_val)
        for k in ("precision", "recall", "ndcg", "hit_ratio"):
            rec["%s.%s" % (nm, k)] = np.asarray(res[k], np.float64)
        rec["%s.users" % nm] = np.array(users, np.int64)
    np.savez_compressed(os.path.join(OUT, "g12_train_trajectory.npz"), **rec)
    args.m_topk_rate, args.T, args.epoch = old
    print("G12: batch losses", [float(rec["b%d.batch_loss" % b]) for b in range(n_batches)])
    print("G12: val recall", rec["val.recall"], "test recall", rec["test.recall"])


# ---------------------------------------------------------------------------------------------------
# G16: the reference loop at the AMAZON-BABY SHAPE (the configuration BASELINE.json's metric is quoted on)
# ---------------------------------------------------------------------------------------------------
BABY = dict(U=35598, I=18357, E=256308, DV=4096, DT=1024, B=1024)
G16_SEED = 3


def digest(t, n=32):
    """(sum, sum of squares, n fixed entries) of a tensor: enough to tell 'the same tensor' without storing it."""
    a = np.asarray(npy(t) if torch.is_tensor(t) else t, np.float64).ravel()
    idx = (np.arange(n, dtype=np.int64) * 2654435761) % max(a.size, 1)
    return np.concatenate([[a.sum(), (a * a).sum()], a[idx]])


def boundary_gaps(ua, ia, users, train_items, Ks):
    """For every tested user: the relative gap between the K-th and (K+1)-th score among non-training items - the margin
    by which its top-K SET is decided. Returns {K: sorted gaps}. A product whose scores differ from the reference's by
    less than a user's gap ranks the same top-K set for that user."""
    out = {k: [] for k in Ks}
    ia_t = ia.t().contiguous()
    for lo in range(0, len(users), 1024):
        ub = users[lo:lo + 1024]
        sc = (ua[ub] @ ia_t).double()
        for r, u in enumerate(ub):
            sc[r, train_items[u]] = -float("inf")
        top = torch.topk(sc, max(Ks) + 1, dim=1).values
        scale = top[:, :1].abs().clamp_min(1e-30)
        for k in Ks:
            out[k].append(((top[:, k - 1] - top[:, k]) / scale[:, 0]).numpy())
    return {k: np.sort(np.concatenate(v)) for k, v in out.items()}


def gen_g16(ref, dg, n_batches=6):
    """G16: the reference's own Trainer.train() for `n_batches` batches + Trainer.test() on EVERY validation and test user
    of a Baby-shaped synthetic dataset (35 598 users x 18 357 items, 256 K interactions, 4096 / 1024-wide features,
    d = 64, B = 1024; k = int(n_items * 1e-4) = 1, T = 1: batches 0-1 on the interaction graph, 2 on the top-1 graph,
    3+ on empty modal graphs). The tensors of this size cannot be committed (the discriminator alone is 2 x 380 MB, one
    batch's Gumbel uniforms 75 MB), so the fixture holds what lets the test REBUILD them and prove it did:
      * the dataset is a pure function of (sizes, seed) (oracle/synth_data.py, numpy Generator);
      * initial parameters and the loop's random tensors come from torch's CPU generator after set_seed(2022): the test
        constructs the product Trainer with its discriminator kept on the CPU while it is initialised (what the shimmed
        reference does here) and draws the noise in the recorded order; digests (sum, sum of squares, 32 entries) of every
        initial tensor and of every noise tensor are stored and compared first;
      * recorded in full: every sampled batch, the per-batch loss components, the final SMALL parameters, 512 sampled
        rows of the final embedding tables / eval-mode embeddings (+ digests of the whole tensors), the metric dicts;
      * the margin by which each tested user's top-K set is decided (relative gap between score K and K + 1)."""
    args = ref.args
    assert (dg.n_users, dg.n_items) == (BABY["U"], BABY["I"]) and args.batch_size == BABY["B"]
    args.epoch = 1
    ref.set_seed(2022)
    tr = ref.Trainer(data_config={"n_users": dg.n_users, "n_items": dg.n_items})
    skip = ("encoder.", "align.", "image_embedding", "text_embedding")
    rec = {"n_batches": n_batches, "k": int(dg.n_items * args.m_topk_rate), "T": int(args.T),
           "m_topk_rate": args.m_topk_rate, "lr": args.lr, "D_lr": args.D_lr, "G_rate": args.G_rate,
           "cl_rate": args.cl_rate, "gp_rate": args.gp_rate, "shape": np.array([dg.n_users, dg.n_items, dg.n_train]),
           "dataset_seed": G16_SEED, "torch_version": str(torch.__version__)}
    assert rec["k"] == 1 and rec["T"] == 1
    for k, v in tr.model.state_dict().items():
        if not k.startswith(skip):
            rec["m0d." + k] = digest(v)
    for k, v in tr.D.state_dict().items():
        if v.dtype.is_floating_point:
            rec["D0d." + k] = digest(v)
    st = {"fwd": 0, "D": 0, "cl": 0, "uni": 0, "alpha": 0, "sample": 0, "on": False}
    o_fwd, o_D, o_bpr, o_cl = tr.model.forward, tr.D.forward, tr.bpr_loss, tr.batched_contrastive_loss
    o_feat, o_gp, o_sample = tr.feat_reg_loss_calculation, tr.gradient_penalty, dg.sample
    o_uniform, o_rand = torch.Tensor.uniform_, torch.rand
    import time
    t0 = time.time()

    def fwd(*graphs):
        c = st["fwd"]
        st["fwd"] += 1
        if st["on"] and c % 2 == 0:
            print("G16: batch %d starts at %.0f s" % (c // 2, time.time() - t0), flush=True)
        if st["on"] and c % 2 == 0 and c // 2 >= n_batches:
            raise _StopTraining()
        return o_fwd(*graphs)

    def D_fwd(x):
        out = o_D(x)
        if st["on"]:
            b, j = divmod(st["D"], 4)
            st["D"] += 1
            rec["b%d.D%d_mean" % (b, j)] = np.float32(out.detach().mean().item())
        return out

    def bpr(u, p, n):
        mf, emb, reg = o_bpr(u, p, n)
        b = st["fwd"] // 2 - 1
        rec["b%d.mf" % b], rec["b%d.emb" % b] = np.float32(mf.item()), np.float32(emb.item())
        return mf, emb, reg

    def cl(z1, z2):
        out = o_cl(z1, z2)
        b, j = divmod(st["cl"], 2)
        st["cl"] += 1
        rec["b%d.cl%d" % (b, j + 1)] = np.float32(out.item())
        return out

    def feat(a, b_, c, d):
        out = o_feat(a, b_, c, d)
        rec["b%d.feat" % (st["fwd"] // 2 - 1)] = np.float32(out.item())
        return out

    def gp(D, xr, xf):
        out = o_gp(D, xr, xf)
        rec["b%d.gp" % (st["fwd"] // 2)] = np.float32(out.item())
        return out

    def sample():
        out = o_sample()
        b = st["sample"]
        st["sample"] += 1
        if b < n_batches:
            rec["b%d.users" % b] = np.array(out[0], np.int32)
            rec["b%d.pos" % b] = np.array(out[1], np.int32)
            rec["b%d.neg" % b] = np.array(out[2], np.int32)
        return out

    def uniform_(self, *a, **k):
        out = o_uniform(self, *a, **k)
        if st["on"]:
            rec["b%d.gumbel_d" % st["uni"]] = digest(out)
            rec["b%d.gumbel_shape" % st["uni"]] = np.array(out.shape)
            st["uni"] += 1
        return out

    def rand(*a, **k):
        out = o_rand(*a, **k)
        if st["on"]:
            rec["b%d.gp_alpha_d" % st["alpha"]] = digest(out)
            st["alpha"] += 1
        return out

    tr.model.forward, tr.D.forward, tr.bpr_loss, tr.batched_contrastive_loss = fwd, D_fwd, bpr, cl
    tr.feat_reg_loss_calculation, tr.gradient_penalty, dg.sample = feat, gp, sample
    torch.Tensor.uniform_, torch.rand = uniform_, rand
    st["on"] = True
    try:
        tr.train()
    except _StopTraining:
        pass
    finally:
        st["on"] = False
        torch.Tensor.uniform_, torch.rand = o_uniform, o_rand
        dg.sample = o_sample
        tr.model.forward, tr.D.forward = o_fwd, o_D
    assert st["uni"] == n_batches and st["alpha"] == n_batches and st["D"] == 4 * n_batches, st
    for b in range(n_batches):
        G_lossf = -float(rec["b%d.D3_mean" % b])
        rec["b%d.G_lossf" % b] = np.float32(G_lossf)
        rec["b%d.loss_D" % b] = np.float32(-float(rec["b%d.D1_mean" % b]) + float(rec["b%d.D0_mean" % b])
                                           + args.gp_rate * float(rec["b%d.gp" % b]))
        rec["b%d.batch_loss" % b] = np.float32(
            float(rec["b%d.mf" % b]) + float(rec["b%d.emb" % b]) + float(rec["b%d.feat" % b])
            + args.cl_rate * (float(rec["b%d.cl1" % b]) + float(rec["b%d.cl2" % b])) + args.G_rate * G_lossf)
    rng = np.random.default_rng(16)
    rows_u = np.sort(rng.choice(dg.n_users, 512, replace=False))
    rows_i = np.sort(rng.choice(dg.n_items, 512, replace=False))
    rec["rows_u"], rec["rows_i"] = rows_u, rows_i
    for k, v in tr.model.state_dict().items():
        if k.startswith(skip):
            continue
        rec["m1d." + k] = digest(v)
        a = npy(v)
        if k == "user_id_embedding.weight":
            rec["m1." + k] = a[rows_u]
        elif k == "item_id_embedding.weight":
            rec["m1." + k] = a[rows_i]
        elif a.size <= 70000:
            rec["m1." + k] = a
        else:                                  # image_trans.weight [64, 4096]: every 8th column
            rec["m1." + k] = a[:, ::8]
    for nm, g in zip(("img_ui", "img_iu", "txt_ui", "txt_iu"),
                     (tr.image_ui_graph, tr.image_iu_graph, tr.text_ui_graph, tr.text_iu_graph)):
        rec["final.%s_nnz" % nm] = int(g._nnz())
    tr.model.eval()
    with torch.no_grad():
        outs = tr.model(tr.ui_graph, tr.iu_graph, tr.image_ui_graph, tr.image_iu_graph, tr.text_ui_graph, tr.text_iu_graph)
    ua, ia = outs[0].detach(), outs[1].detach()
    rec["eval.ua"], rec["eval.ia"] = npy(ua)[rows_u], npy(ia)[rows_i]
    rec["eval.ua_d"], rec["eval.ia_d"] = digest(ua), digest(ia)
    Ks = eval(args.Ks)
    for is_val, nm in ((True, "val"), (False, "test")):
        users = [u for u, v in (dg.val_set if is_val else dg.test_set).items() if len(v) > 0]
        print("G16: evaluating %d %s users at %.0f s" % (len(users), nm, time.time() - t0), flush=True)
        res = tr.test(users, is_val)
        for k in ("precision", "recall", "ndcg", "hit_ratio"):
            rec["%s.%s" % (nm, k)] = np.asarray(res[k], np.float64)
        rec["%s.n_users" % nm] = len(users)
        rec["%s.users_d" % nm] = digest(np.array(users, np.float64))
    # how firmly each tested user's top-K set is decided (test users; training items masked as test_one_user does)
    users = [u for u, v in dg.test_set.items() if len(v) > 0]
    gaps = boundary_gaps(ua, ia, users, dg.train_items, Ks)
    for k in Ks:
        rec["test.gap%d_smallest" % k] = gaps[k][:64]
        rec["test.gap%d_below_1e-5" % k] = int((gaps[k] < 1e-5).sum())
        rec["test.gap%d_below_1e-4" % k] = int((gaps[k] < 1e-4).sum())
    np.savez_compressed(os.path.join(OUT, "g16_baby_trajectory.npz"), **rec)
    print("G16: batch losses", [float(rec["b%d.batch_loss" % b]) for b in range(n_batches)])
    print("G16: val recall", rec["val.recall"], "test recall", rec["test.recall"])
    print("G16: users whose top-20 set is decided by < 1e-5 / 1e-4 of the top score:", rec["test.gap20_below_1e-5"],
          rec["test.gap20_below_1e-4"], "of", len(users))
    print("G16: %d bytes" % os.path.getsize(os.path.join(OUT, "g16_baby_trajectory.npz")))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "g12":     # own process: discriminator dropout must be 0 at import time
        if os.path.isdir(TMP):
            shutil.rmtree(TMP)
        synth_data.write_dataset(TMP, "tiny", U, I, E, DV, DT, seed=1)
        _ref = ref_shim.load(TMP, "tiny", ["--batch_size", str(B), "--drop_rate", "0.0", "--G_drop1", "0.0",
                                           "--G_drop2", "0.0"])
        gen_g12(_ref, _ref.data_generator)
    elif len(sys.argv) > 1 and sys.argv[1] == "g16":     # the Baby shape: ~15 min of CPU, ~12 GB
        TMP16 = "/tmp/mmssl_golden_baby/"
        if not os.path.isfile(TMP16 + "baby/text_feat.npy"):
            synth_data.write_dataset(TMP16, "baby", BABY["U"], BABY["I"], BABY["E"], BABY["DV"], BABY["DT"], seed=G16_SEED)
        _ref = ref_shim.load(TMP16, "baby", ["--batch_size", str(BABY["B"]), "--drop_rate", "0.0", "--G_drop1", "0.0",
                                             "--G_drop2", "0.0"])
        gen_g16(_ref, _ref.data_generator)
    elif len(sys.argv) > 1 and sys.argv[1] == "g9":      # only (re)generate G9; same data set, same seeds
        if os.path.isdir(TMP):
            shutil.rmtree(TMP)
        synth_data.write_dataset(TMP, "tiny", U, I, E, DV, DT, seed=1)
        _ref = ref_shim.load(TMP, "tiny", ["--batch_size", str(B), "--drop_rate", "0.0"])
        gen_g9(_ref, _ref.data_generator)
    else:
        main()
