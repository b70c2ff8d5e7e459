"""`Trainer` / `set_seed` with the reference's interface and training-loop semantics
(/root/reference/MMSSL/main.py:37-536), re-built around the HIP hot path:

  * graphs are `GraphPlan`s (device CSR + transpose + balanced work list) instead of torch COO;
  * `model(...)`, `bpr_loss`, `batched_contrastive_loss`, `feat_reg_loss_calculation` run on the
    kernels of libmmssl_hip.so (see mmssl_amd/ops.py);
  * the adversarial pieces (u_sim_calculation, Discriminator, gradient penalty, Gumbel noise)
    are adjacent to the hot path and stay on stock PyTorch-ROCm ops.

Loop quirks are reproduced, not fixed: two model forwards per batch, modal
graphs rebuilt every T batches from the previous batch's top-k and EMPTY from the third batch
on under the defaults, LambdaLR created but never stepped, AdamW over all model parameters.
Not reproduced: the two dense U x I GPU copies of the train matrix that the reference allocates
and never reads (main.py:59-60), dead methods that reference undefined attributes / dgl
(main.py:114-209, 260-279).
"""
import math
import os
import pickle
import random
import sys
from datetime import datetime
from time import time

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim
from torch import autograd

from . import ops
from .config import args, configure
from .graph import GraphPlan
from .optim import FusedAdamW
from .Models import MMSSL, Discriminator
from .utility import batch_test
from .utility.logging import Logger


def set_seed(seed):
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    ops.seed_dropout(seed)                 # the dropout keep-mask generator of libmmssl_hip (Models.py:54)


class Trainer(object):
    def __init__(self, data_config):
        self.task_name = "%s_%s_%s" % (datetime.now().strftime("%Y-%m-%d %H:%M:%S"), args.dataset, args.cf_model)
        self.logger = Logger(filename=self.task_name, is_debug=args.debug)
        self.logger.logging("PID: %d" % os.getpid())
        self.logger.logging(str(args))

        self.mess_dropout = eval(args.mess_dropout)
        self.lr = args.lr
        self.emb_dim = args.embed_size
        self.batch_size = args.batch_size
        self.weight_size = eval(args.weight_size)
        self.n_layers = len(self.weight_size)
        self.regs = eval(args.regs)
        self.decay = self.regs[0]

        base = args.data_path + args.dataset
        self.image_feats = np.load(base + "/image_feat.npy")
        self.text_feats = np.load(base + "/text_feat.npy")
        self.image_feat_dim = self.image_feats.shape[-1]
        self.text_feat_dim = self.text_feats.shape[-1]
        with open(base + "/train_mat", "rb") as f:
            self.ui_graph_raw = pickle.load(f).tocsr()
        self.image_ui_index = {"x": [], "y": []}
        self.text_ui_index = {"x": [], "y": []}
        self.n_users, self.n_items = self.ui_graph_raw.shape
        self.device = torch.device("cuda", torch.cuda.current_device())

        self.ui_graph = self.matrix_to_tensor(self.csr_norm(self.ui_graph_raw, mean_flag=True))
        self.iu_graph = self.matrix_to_tensor(self.csr_norm(self.ui_graph_raw.T, mean_flag=True))
        self.image_ui_graph = self.text_ui_graph = self.ui_graph
        self.image_iu_graph = self.text_iu_graph = self.iu_graph

        self.model = MMSSL(self.n_users, self.n_items, self.emb_dim, self.weight_size, self.mess_dropout,
                           self.image_feats, self.text_feats).cuda()
        self.D = Discriminator(self.n_items).cuda()
        self.D.apply(self.weights_init)
        self.optim_D = optim.Adam(self.D.parameters(), lr=args.D_lr, betas=(0.5, 0.9))
        # the generator's AdamW (named optimizer_D in the reference, main.py:76-80): one-launch fused kernel
        self.optimizer_D = FusedAdamW([{"params": self.model.parameters()}], lr=self.lr)
        self.scheduler_D = self.set_lr_scheduler()
        self._idx_cache = (None, None)
        self._empty_plans = None
        # Parity runs: `noise_hook(kind, shape)` -> CPU float tensor replaces the two random draws of a batch
        # ("gumbel": the uniforms of main.py:350, "gp_alpha": torch.rand of main.py:147). None = device generator.
        self.noise_hook = None
        if batch_test.data_generator is None:
            batch_test.init_data()
        self.data_generator = batch_test.data_generator

    # ---- pieces with the reference's names ----------------------------------------------------
    def set_lr_scheduler(self):
        return optim.lr_scheduler.LambdaLR(self.optimizer_D, lr_lambda=lambda epoch: 0.96 ** (epoch / 50))

    def csr_norm(self, csr_mat, mean_flag=False):
        """diag((rowsum+1e-8)^-1/2) . A [. diag((colsum+1e-8)^-1/2)]  (main.py:89-103). Host/scipy,
        like the reference; with mean_flag=True an edge (r, c) becomes 1/sqrt(deg(r))."""
        def inv_sqrt(v):
            v = np.power(np.asarray(v).flatten() + 1e-8, -0.5)
            v[np.isinf(v)] = 0.0
            return sp.diags(v)
        left = inv_sqrt(csr_mat.sum(1))
        if mean_flag:
            return left * csr_mat
        return left * csr_mat * inv_sqrt(csr_mat.sum(0))

    def matrix_to_tensor(self, cur_matrix):
        """scipy matrix -> device graph handle (the reference returns a torch COO tensor here; same one-argument
        signature). While the modal graphs are rebuilt (`_plan_xcd_bands` = -1) the plan skips its XCD-band / co-clustering
        pass: not worth it for a plan that lives one batch."""
        return GraphPlan(cur_matrix, xcd_bands=getattr(self, "_plan_xcd_bands", 0))

    def sparse_mx_to_torch_sparse_tensor(self, sparse_mx):
        return GraphPlan(sparse_mx)

    def weights_init(self, m):
        if isinstance(m, nn.Linear):
            nn.init.kaiming_normal_(m.weight)
            m.bias.data.fill_(0)

    def gradient_penalty(self, D, xr, xf):
        lam = 0.3
        xf, xr = xf.detach(), xr.detach()
        if self.noise_hook is not None:
            alpha = self.noise_hook("gp_alpha", (args.batch_size * 2, 1)).to(xr.device)
        else:
            alpha = torch.rand(args.batch_size * 2, 1, device=xr.device)
        alpha = alpha.expand_as(xr)
        inter = (alpha * xr + (1 - alpha) * xf).requires_grad_()
        out = D(inter)
        grads = autograd.grad(outputs=out, inputs=inter, grad_outputs=torch.ones_like(out), create_graph=True,
                              retain_graph=True, only_inputs=True)[0]
        return ((grads.norm(2, dim=1) - 1) ** 2).mean() * lam

    def sim(self, z1, z2):
        return torch.mm(ops.l2norm_rows(z1), ops.l2norm_rows(z2).t())

    def batched_contrastive_loss(self, z1, z2, batch_size=1024):
        """main.py:218-249 in one fused kernel sequence; `batch_size` only shaped the reference's
        memory blocking (the result is the full-matrix formula) and is ignored."""
        return ops.infonce(z1, z2, args.tau)

    def feat_reg_loss_calculation(self, g_item_image, g_item_text, g_user_image, g_user_text):
        feat_reg = 0.5 * self.model.feat_sumsq(g_item_image, g_item_text, g_user_image, g_user_text)
        return args.feat_reg_decay * (feat_reg / self.n_items)

    def bpr_loss(self, users, pos_items, neg_items):
        """(mf_loss, emb_loss, reg_loss) from already-gathered [B, d] rows (main.py:499-511)."""
        mf_loss, emb_loss = ops.bpr(users, pos_items, neg_items, self.decay, self.batch_size)
        return mf_loss, emb_loss, 0.0

    def _batch_idx(self, users, pos_items=None, neg_items=None):
        """int64 device tensors of one sampled batch, uploaded ONCE per batch object through pinned memory
        (non-blocking: the host does not wait for the kernels queued before the copy). Returns (u, p, n);
        p / n are None until a call supplies the item lists."""
        key, val = self._idx_cache
        if key is not users:
            val = [self._upload(users), None, None]
            self._idx_cache = (users, val)
        if pos_items is not None and val[1] is None:
            val[1], val[2] = self._upload(pos_items), self._upload(neg_items)
        return val

    def _upload(self, ids):
        host = torch.from_numpy(np.asarray(ids, dtype=np.int64)).pin_memory()
        return host.to(self.device, non_blocking=True)

    def _users_idx(self, users):
        return self._batch_idx(users)[0]

    def _seen_rows(self, users):
        """Dense 0/1 rows of the train matrix for this batch, built ON THE DEVICE from the CSR pattern of
        the user-item plan (the reference calls ui_graph_raw[users].todense() and uploads [B, n_items]
        floats in every u_sim_calculation call and once more for the Gumbel step, main.py:283,349)."""
        return ops.graph_rows_dense(self.ui_graph, self._users_idx(users), 1.0)

    def u_sim_calculation(self, users, user_final, item_final):
        """main.py:281-298: library GEMM + fused mask/normalise over the plan's CSR rows (ops.usim)."""
        return ops.usim(self._users_idx(users), user_final, item_final, self.ui_graph)

    def _graphs(self):
        return (self.ui_graph, self.iu_graph, self.image_ui_graph, self.image_iu_graph, self.text_ui_graph,
                self.text_iu_graph)

    def test(self, users_to_test, is_val):
        self.model.eval()
        with torch.no_grad():
            ua_embeddings, ia_embeddings, *rest = self.model(*self._graphs())
        return batch_test.test_torch(ua_embeddings, ia_embeddings, users_to_test, is_val, data=self.data_generator)

    # ---- one batch ----------------------------------------------------------------------------
    def _discriminator_step(self, users, outs=None):
        if outs is None:
            with torch.no_grad():
                outs = self.model(*self._graphs())
        ua, ia, img_item, txt_item, img_user, txt_user = outs[:6]
        ui_u_sim = self.u_sim_calculation(users, ua, ia).detach()
        inputf = torch.cat((self.u_sim_calculation(users, img_user, img_item).detach(),
                            self.u_sim_calculation(users, txt_user, txt_item).detach()), dim=0)
        lossf = self.D(inputf).mean()
        u_ui = self._seen_rows(users)
        if self.noise_hook is not None:
            noise = self.noise_hook("gumbel", tuple(u_ui.shape)).to(u_ui.device)
        elif os.environ.get("MMSSL_REF_NOISE", "0") == "1":
            # the reference draws the Gumbel noise from the CPU generator and uploads it (main.py:350): with the same
            # set_seed the discriminator then sees the reference's noise stream (costs a [B, n_items] host draw + copy)
            noise = torch.empty(u_ui.shape, dtype=torch.float32).uniform_(0, 1).pin_memory().to(u_ui.device, non_blocking=True)
        else:
            noise = torch.empty_like(u_ui).uniform_(0, 1)
        u_ui = F.softmax(u_ui - args.log_log_scale * torch.log(-torch.log(noise + 1e-8) + 1e-8) / args.real_data_tau,
                         dim=1)
        u_ui = F.normalize(u_ui + ui_u_sim * args.ui_pre_scale, dim=1)
        inputr = torch.cat((u_ui, u_ui), dim=0)
        lossr = -self.D(inputr).mean()
        gp = self.gradient_penalty(self.D, inputr, inputf.detach())
        loss_D = lossr + lossf + args.gp_rate * gp
        self.optim_D.zero_grad()
        loss_D.backward()
        self.optim_D.step()
        return loss_D.detach()

    def _maintain_modal_graphs_device(self, idx, users, img_sim, txt_sim, k):
        """The same bookkeeping as _maintain_modal_graphs with everything on the device: top-k by the selection
        kernel (ops.topk_rows), the (user, item) pair lists as device tensors, the four graphs rebuilt in place by
        graph.DeviceGraphPair (CSR + transpose + work lists built by kernels; no .cpu(), no python lists, no scipy,
        no plan upload). Used when the collected pairs fit its capacity (MMSSL_DEVICE_GRAPHS=0 disables it)."""
        from .graph import DeviceGraphPair
        store = self.__dict__.setdefault("_dev_pairs", {"image": [], "text": []})
        if idx % args.T == 0 and idx != 0:
            for name in ("image", "text"):
                if not store[name]:             # nothing collected: the cached empty plans (no launches at all)
                    self._set_empty_modal(name)
                    continue
                pair = self.__dict__.setdefault("_dev_graphs", {}).get(name)
                if pair is None:
                    pair = DeviceGraphPair(self.n_users, self.n_items, DeviceGraphPair.MAX_PAIRS)
                    self._dev_graphs[name] = pair
                xs = [x for x, _ in store[name]]
                ys = [y for _, y in store[name]]
                empty = torch.empty(0, dtype=torch.int64, device=self.device)
                pair.rebuild(torch.cat(xs) if xs else empty, torch.cat(ys) if ys else empty)
                setattr(self, name + "_ui_graph", pair.ui)
                setattr(self, name + "_iu_graph", pair.iu)
            self._dev_pairs = {"image": [], "text": []}
        else:
            u = self._users_idx(users)
            for name, s in (("image", img_sim), ("text", txt_sim)):
                ids = ops.topk_rows(s, k)                       # [B, k], descending score, lowest id on ties
                # the reference pairs the user list TILED k times with the row-major top-k ids (tensor.repeat(1, k))
                store[name].append((u.repeat(k), ids.reshape(-1)))

    def _set_empty_modal(self, name):
        if self._empty_plans is None:
            e = sp.csr_matrix((self.n_users, self.n_items), dtype=np.float32)
            self._empty_plans = (self.matrix_to_tensor(e), self.matrix_to_tensor(e.T.tocsr()))
        setattr(self, name + "_ui_graph", self._empty_plans[0])
        setattr(self, name + "_iu_graph", self._empty_plans[1])

    def _device_graphs_ok(self, k, n_batch_users):
        from .graph import DeviceGraphPair
        if os.environ.get("MMSSL_DEVICE_GRAPHS", "1") == "0" or k < 1 or k > ops.TOPK_MAX_K:
            return False
        # collected pairs carry over the epoch boundary like the reference's lists (cleared only by a rebuild): up to T-1
        # batches are left after an epoch's last rebuild and idx 0..T-1 of the next epoch add T more
        worst = (2 * max(int(args.T), 1) - 1) * n_batch_users * k
        # (catalogues wider than one top-k launch are ranked block by block inside ops.topk_rows)
        return worst <= DeviceGraphPair.MAX_PAIRS and k * (-(-self.n_items // ops.TOPK_MAX_COLS)) <= ops.TOPK_MAX_COLS

    def _maintain_modal_graphs(self, idx, users, img_sim, txt_sim):
        """main.py:378-405: every T-th batch (idx != 0) rebuild the four modal graphs from the collected (user, top-k
        item) pairs and clear the lists; otherwise collect. On the device when the pairs fit (see
        _maintain_modal_graphs_device), else on the host like the reference (scipy)."""
        k = int(self.n_items * args.m_topk_rate)
        if img_sim.is_cuda and self._device_graphs_ok(k, len(users)):
            return self._maintain_modal_graphs_device(idx, users, img_sim, txt_sim, k)
        if idx % args.T == 0 and idx != 0:
            shape = (self.n_users, self.n_items)
            for name, store in (("image", self.image_ui_index), ("text", self.text_ui_index)):
                if not store["x"]:
                    # nothing collected (the reference's steady state from the third batch on):
                    # the rebuilt graphs are empty; reuse one empty plan pair instead of 4 scipy + plan builds
                    self._set_empty_modal(name)
                    continue
                tmp = sp.csr_matrix((np.ones(len(store["x"]), np.float32), (store["x"], store["y"])), shape=shape)
                self._plan_xcd_bands = -1
                try:
                    setattr(self, name + "_ui_graph", self.matrix_to_tensor(self.csr_norm(tmp, mean_flag=True)))
                    setattr(self, name + "_iu_graph", self.matrix_to_tensor(self.csr_norm(tmp.T, mean_flag=True)))
                finally:
                    self._plan_xcd_bands = 0
            self.image_ui_index = {"x": [], "y": []}
            self.text_ui_index = {"x": [], "y": []}
        else:
            rep = np.repeat(np.asarray(users, dtype=np.int64)[None, :], k, axis=0).reshape(-1).tolist() if k else []
            # the reference tiles the whole user list k times (tensor.repeat(1, k)), not each user k times
            for store, s in ((self.image_ui_index, img_sim), (self.text_ui_index, txt_sim)):
                if k:
                    _, ids = torch.topk(s, k, dim=-1)
                    store["x"] += rep
                    store["y"] += ids.cpu().view(-1).tolist()

    def generator_losses(self, idx, users, pos_items, neg_items, keep_masks=None, maintain_graphs=True):
        """The generator-step loss assembly (main.py:363-420) without the optimiser step."""
        (G_ua, G_ia, G_img_item, G_txt_item, G_img_user, G_txt_user, G_user_emb, _, G_img_uid, G_txt_uid, _, _) = \
            self.model(*self._graphs(), keep_masks=keep_masks)
        dev = self.device
        u_idx, p_idx, n_idx = self._batch_idx(users, pos_items, neg_items)
        # gathers + bpr_loss (main.py:368-371) and both batched_contrastive_loss calls (:411-412) as ONE
        # autograd node on the full tables (G_user_emb is G_ua: outputs 0 and 6 of the model are one tensor)
        assert G_user_emb is G_ua
        mf_loss, emb_loss, cl1, cl2 = ops.batch_losses(G_ua, G_ia, G_img_uid, G_txt_uid, u_idx, p_idx, n_idx,
                                                       self.decay, self.batch_size, args.tau)
        reg_loss = 0.0
        G_img_sim = self.u_sim_calculation(users, G_img_user, G_img_item)
        G_txt_sim = self.u_sim_calculation(users, G_txt_user, G_txt_item)
        if maintain_graphs:
            self._maintain_modal_graphs(idx, users, G_img_sim.detach(), G_txt_sim.detach())
        feat_emb_loss = self.feat_reg_loss_calculation(G_img_item, G_txt_item, G_img_user, G_txt_user)
        # The generator loss needs d(D)/d(input) only. The reference also accumulates this step's gradients
        # into D's own parameters, which nothing reads (optim_D.zero_grad() clears them before D's next
        # backward, main.py:357-359): with D's parameters frozen while the graph is built, autograd skips
        # those [2B, n_items] x [n_items, n_items/4] weight-gradient GEMMs.
        d_params = [p for p in self.D.parameters() if p.requires_grad]
        for p in d_params:
            p.requires_grad_(False)
        try:
            G_lossf = -(self.D(torch.cat((G_img_sim, G_txt_sim), dim=0)).mean())
        finally:
            for p in d_params:
                p.requires_grad_(True)
        batch_loss = mf_loss + emb_loss + reg_loss + feat_emb_loss + args.cl_rate * (cl1 + cl2) \
            + args.G_rate * G_lossf
        return dict(batch_loss=batch_loss, mf=mf_loss, emb=emb_loss, reg=reg_loss, feat=feat_emb_loss, cl1=cl1,
                    cl2=cl2, G_lossf=G_lossf)

    def _generator_step(self, idx, users, pos_items, neg_items):
        L = self.generator_losses(idx, users, pos_items, neg_items)
        self.optimizer_D.zero_grad()
        L["batch_loss"].backward(retain_graph=False)
        self.optimizer_D.step()
        # detached: a returned loss that still carried its autograd graph would keep the parameters' gradient
        # accumulators (bound to THIS stream) alive into a later hipGraph capture on another stream
        det = lambda t: t.detach() if torch.is_tensor(t) else t      # noqa: E731
        return (det(L["batch_loss"]), det(L["mf"]), det(L["emb"]), L["reg"], det(L["cl1"] + L["cl2"]), det(L["G_lossf"]))

    # ---- the same batch on the captured hot path ---------------------------------------------------
    def _steady_state(self):
        """True once the four modal graphs are the cached empty plans and nothing is being collected — the
        reference's state from the fourth batch on under its defaults — i.e. when the graph handles
        no longer change from batch to batch and a captured step stays valid."""
        e = self._empty_plans
        # with k == 0 nothing is ever collected; with T == 1 every batch from the second on takes the rebuild branch,
        # so nothing is collected after the first batch either (Baby: k = 1, T = 1): the graphs stay empty for good
        never_collects = int(self.n_items * args.m_topk_rate) == 0 or int(args.T) == 1
        return (e is not None and self.image_ui_graph is e[0] and self.text_ui_graph is e[0]
                and self.image_iu_graph is e[1] and self.text_iu_graph is e[1] and never_collects)

    def _captured(self):
        """SplitHotPath for the current (steady) graph set, captured on first use; None if disabled / refused."""
        if os.environ.get("MMSSL_TRAINER_GRAPH", "1") == "0" or not self._steady_state():
            return None
        cap = getattr(self, "_split", None)
        if cap is None:
            # the first steady-state batch still runs op by op: everything a steady-state step allocates on first
            # use (cached zero views of the empty modal graphs, workspaces of the empty plans) then exists before the
            # capture
            self._steady_seen = getattr(self, "_steady_seen", 0) + 1
            if self._steady_seen < 2:
                return None
            from .hotpath import SplitHotPath
            cap = SplitHotPath(self.model, self._graphs(), self.optimizer_D, self.batch_size, self.decay,
                               [1.0, 1.0, 1.0, args.cl_rate, args.cl_rate], args.feat_reg_decay * 0.5 / self.n_items)
            if not cap.capture():
                cap = False
            self._split = cap
        return cap or None

    def train_batch(self, idx, users, pos_items, neg_items):
        """Discriminator step + generator step of one batch (main.py:334-429): on the captured hot path once the
        graph handles are stable, op by op before that (and when MMSSL_TRAINER_GRAPH=0)."""
        cap = self._captured()
        if cap is not None:
            return self._batch_captured(cap, idx, users, pos_items, neg_items)
        self._discriminator_step(users)
        return self._generator_step(idx, users, pos_items, neg_items)

    def _batch_captured(self, cap, idx, users, pos_items, neg_items):
        """One batch of main.py:334-429 with the hot path replayed from two hipGraphs (hotpath.SplitHotPath): the
        no-grad forward of the discriminator step and the generator's forward are replays of segment F, the
        generator's backward + AdamW is segment B; u_sim / Discriminator / gradient penalty run eagerly in between
        and hand their gradient w.r.t. the modal feature outputs to B."""
        u_idx, p_idx, n_idx = self._batch_idx(users, pos_items, neg_items)
        batch3 = torch.stack((u_idx, p_idx, n_idx))
        o = cap.forward(batch3)                                   # main.py:340-343 (dropout active, no grad needed)
        self._discriminator_step(users, outs=tuple(t.detach() for t in o[:6]))
        o = cap.forward(batch3)                                   # main.py:363-365
        leaves = [o[k].detach().requires_grad_(True) for k in (2, 3, 4, 5)]      # img_item, txt_item, img_user, txt_user
        G_img_sim = self.u_sim_calculation(users, leaves[2], leaves[0])
        G_txt_sim = self.u_sim_calculation(users, leaves[3], leaves[1])
        self._maintain_modal_graphs(idx, users, G_img_sim.detach(), G_txt_sim.detach())
        d_params = [p for p in self.D.parameters() if p.requires_grad]
        for p in d_params:
            p.requires_grad_(False)
        try:
            G_lossf = -(self.D(torch.cat((G_img_sim, G_txt_sim), dim=0)).mean())
        finally:
            for p in d_params:
                p.requires_grad_(True)
        (args.G_rate * G_lossf).backward()
        cap.backward([t.grad for t in leaves])
        terms = cap.terms
        batch_loss = cap.loss + args.G_rate * G_lossf.detach()
        return batch_loss, terms[0], terms[1], 0.0, terms[3] + terms[4], G_lossf.detach()

    # ---- training loop --------------------------------------------------------------------------
    def train(self):
        run_time = datetime.strftime(datetime.now(), "%Y_%m_%d__%H_%M_%S")
        dg = self.data_generator
        stopping_step, best_recall, test_ret = 0, 0, None
        Ks = eval(args.Ks)
        nxt = dg.sample()
        for epoch in range(args.epoch):
            t1 = time()
            loss = mf_loss = emb_loss = reg_loss = 0.0
            n_batch = dg.n_train // args.batch_size + 1
            for idx in range(n_batch):
                self.model.train()
                users, pos_items, neg_items = nxt
                self._batch_idx(users, pos_items, neg_items)       # uploads before any kernel of the batch
                bl, mf, emb, reg, _, _ = self.train_batch(idx, users, pos_items, neg_items)
                # the next batch is sampled while the device works on this one: Data.sample() consumes the host
                # RNG streams in exactly the reference's order (one call per batch, nothing else draws from them)
                nxt = dg.sample()
                loss += float(bl)
                mf_loss += float(mf)
                emb_loss += float(emb)
                reg_loss += float(reg)
            if math.isnan(loss):
                self.logger.logging("ERROR: loss is nan.")
                sys.exit()
            t2 = time()
            if args.verbose and (epoch + 1) % args.verbose != 0:      # main.py:444-448 (evaluation still follows)
                self.logger.logging("Epoch %d [%.1fs]: train==[%.5f=%.5f + %.5f + %.5f]" % (
                    epoch, t2 - t1, loss, mf_loss, emb_loss, reg_loss))
            ret = self.test(list(dg.val_set.keys()), is_val=True)
            t3 = time()
            if args.verbose > 0:
                self.logger.logging(
                    "Epoch %d [%.1fs + %.1fs]: train==[%.5f=%.5f + %.5f + %.5f], recall=[%s], precision=[%s], "
                    "hit=[%s], ndcg=[%s]" % (epoch, t2 - t1, t3 - t2, loss, mf_loss, emb_loss, reg_loss,
                                             *(", ".join("%.5f" % v for v in ret[k])
                                               for k in ("recall", "precision", "hit_ratio", "ndcg"))))
            if ret["recall"][1] > best_recall:
                best_recall = ret["recall"][1]
                test_ret = self.test(list(dg.test_set.keys()), is_val=False)
                self.logger.logging("Test_Recall@%d: %.5f,  precision=[%.5f], ndcg=[%.5f]" % (
                    Ks[1], test_ret["recall"][1], test_ret["precision"][1], test_ret["ndcg"][1]))
                stopping_step = 0
            elif stopping_step < args.early_stopping_patience:
                stopping_step += 1
                self.logger.logging("#####Early stopping steps: %d #####" % stopping_step)
            else:
                self.logger.logging("#####Early stop! #####")
                break
        self.logger.logging(str(test_ret))
        return best_recall, run_time


def main(argv=None):
    configure(sys.argv[1:] if argv is None else argv)
    torch.cuda.set_device(args.gpu_id)
    set_seed(args.seed)
    dg = batch_test.init_data()
    trainer = Trainer(data_config={"n_users": dg.n_users, "n_items": dg.n_items})
    return trainer.train()


if __name__ == "__main__":
    main()
