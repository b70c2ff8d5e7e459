"""Training loops of the two baselines the reference ships next to MMSSL (SURVEY.md 8f "next #4"):
/root/reference/LATTICE/codes/main.py:23-185 and /root/reference/MICRO/codes/main.py:24-190 — same `Trainer` surface
(`__init__(data_config)`, `set_lr_scheduler`, `test`, `train`, `bpr_loss`), same order of random draws and the same
arithmetic per batch, on the HIP models of `mmssl_amd/baselines.py`:

    epoch:  for every batch   users, pos, neg = data_generator.sample()
                              embeddings      = model(norm_adj, build_item_graph = first batch of the epoch)
                              loss            = BPR (+ loss_ratio * (InfoNCE(image, h) + InfoNCE(text, h)) for MICRO)
                              loss.backward(); Adam step
            lr *= 0.96 ** (1 / 50) per epoch (LambdaLR); every `verbose` epochs validation, test on a new best Recall@20,
            early stopping on `early_stopping_patience` validations without one.

What differs from the reference is where things run: the (U+I)^2 adjacency is a `GraphPlan` instead of a torch COO tensor,
the batch rows are gathered inside the BPR kernel (`ops.bpr_gather`) instead of three index_selects, `optim.Adam` is
`FusedAdamW(weight_decay=0)` — the same update rule — in one launch, and the evaluation is the product's `test_torch`.
`parse_args(model_name)` mirrors the two parsers (LATTICE/codes/utility/parser.py, MICRO/codes/utility/parser.py)."""
import argparse
import math
from time import time

import numpy as np
import torch

from . import baselines, config, ops
from .graph import GraphPlan
from .optim import FusedAdamW
from .utility import batch_test


def parse_args(model_name="lattice", argv=None):
    m = model_name.lower()
    if m not in ("lattice", "micro"):
        raise ValueError("model_name %r" % (model_name,))
    p = argparse.ArgumentParser(description="")
    p.add_argument("--data_path", nargs="?", default="../data")
    p.add_argument("--seed", type=int, default=123)
    p.add_argument("--dataset", nargs="?", default="cloth")
    p.add_argument("--verbose", type=int, default=5)
    p.add_argument("--epoch", type=int, default=200 if m == "lattice" else 1000)
    p.add_argument("--batch_size", type=int, default=1024)
    p.add_argument("--regs", nargs="?", default="[1e-5,1e-5,1e-2]")
    p.add_argument("--lr", type=float, default=0.0005)
    p.add_argument("--embed_size", type=int, default=64)
    p.add_argument("--weight_size", nargs="?", default="[64,64]")
    p.add_argument("--core", type=int, default=5)
    p.add_argument("--topk", type=int, default=10)
    p.add_argument("--lambda_coeff", type=float, default=0.9)
    p.add_argument("--cf_model", nargs="?", default="lightgcn")
    p.add_argument("--mess_dropout", nargs="?", default="[0.1, 0.1]")
    p.add_argument("--early_stopping_patience", type=int, default=10)
    p.add_argument("--gpu_id", type=int, default=0 if m == "lattice" else 1)
    p.add_argument("--Ks", nargs="?", default="[10, 20, 50]" if m == "lattice" else "[10, 20]")
    p.add_argument("--test_flag", nargs="?", default="part")
    if m == "lattice":
        p.add_argument("--model_name", nargs="?", default="lattice")
        p.add_argument("--feat_embed_dim", type=int, default=64)
        p.add_argument("--n_layers", type=int, default=1)
    else:
        p.add_argument("--layers", type=int, default=1)
        p.add_argument("--sparse", type=int, default=1)
        p.add_argument("--debug", action="store_true")
        p.add_argument("--loss_ratio", type=float, default=0.03)
        p.add_argument("--norm_type", nargs="?", default="sym")
    a = p.parse_args(argv)
    a.model_name = m
    return a


def set_seed(seed):
    import random
    np.random.seed(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class Trainer(object):
    def __init__(self, data_config, args, data=None, device=None):
        self.args = args
        self.data = data if data is not None else batch_test.data_generator
        if self.data is None:
            raise RuntimeError("baselines_main.Trainer: batch_test.init_data() first, or pass data=")
        self.device = torch.device(device or "cuda")
        self.n_users = data_config["n_users"]
        self.n_items = data_config["n_items"]
        self.model_name = args.model_name
        self.mess_dropout = eval(args.mess_dropout)
        self.lr = args.lr
        self.emb_dim = args.embed_size
        self.batch_size = args.batch_size
        self.weight_size = eval(args.weight_size)
        self.n_layers = len(self.weight_size)
        self.regs = eval(args.regs)
        self.decay = self.regs[0]
        self.norm_adj = self._make_adj(data_config["norm_adj"])
        image_feats = np.load(args.data_path + "{}/image_feat.npy".format(args.dataset))
        text_feats = np.load(args.data_path + "{}/text_feat.npy".format(args.dataset))
        self.model = self._make_model(image_feats, text_feats).to(self.device)
        self.optimizer = self._make_optimizer()
        self.lr_scheduler = self.set_lr_scheduler()
        self.history = []          # one dict per validation: what the reference prints

    # ---- the three places a test may substitute (reference classes on CPU: oracle/gen_golden_baselines.py) ----------
    def _make_adj(self, norm_adj):
        return GraphPlan(norm_adj.tocsr().astype(np.float32))

    def _make_model(self, image_feats, text_feats):
        a = self.args
        common = dict(topk=a.topk, lambda_coeff=a.lambda_coeff, cf_model=a.cf_model)
        if self.model_name == "lattice":
            return baselines.LATTICE(self.n_users, self.n_items, self.emb_dim, self.weight_size, self.mess_dropout,
                                     image_feats, text_feats, feat_embed_dim=a.feat_embed_dim, n_layers=a.n_layers, **common)
        return baselines.MICRO(self.n_users, self.n_items, self.emb_dim, self.weight_size, self.mess_dropout,
                               image_feats, text_feats, layers=a.layers, norm_type=a.norm_type, **common)

    # parameters that shape the LEARNED item graph get a gradient only in the batch that rebuilds it (the first of an
    # epoch; afterwards the graph is detached): optim.Adam counts steps per parameter and skips a parameter without a
    # gradient, the fused optimiser counts per GROUP and skips a group without gradients - so they form a group of their own
    GRAPH_PARAMS = ("image_embedding.", "text_embedding.", "image_trs.", "text_trs.", "modal_weight")

    def _make_optimizer(self):
        graph, rest = [], []
        for name, p in self.model.named_parameters():
            (graph if name.startswith(self.GRAPH_PARAMS) else rest).append(p)
        return FusedAdamW([{"params": rest}, {"params": graph}], lr=self.lr, weight_decay=0.0)    # optim.Adam(lr) (main.py:47)

    def _evaluate(self, ua, ia, users_to_test, is_val):
        cfg = config.args
        keep = (cfg.Ks, cfg.batch_size)
        cfg.Ks, cfg.batch_size = self.args.Ks, self.args.batch_size
        try:
            return batch_test.test_torch(ua, ia, users_to_test, is_val, data=self.data)
        finally:
            cfg.Ks, cfg.batch_size = keep

    def _batch_losses(self, outs, users, pos_items, neg_items):
        """(mf, emb, reg, contrastive) of one batch from the model's outputs."""
        mf, emb = ops.bpr_gather(outs[0], outs[1], users, pos_items, neg_items, self.decay, self.batch_size)
        cl = None
        if self.model_name == "micro":
            cl = self.model.batched_contrastive_loss(outs[2], outs[4]) + self.model.batched_contrastive_loss(outs[3], outs[4])
            cl = cl * self.args.loss_ratio
        return mf, emb, 0.0, cl

    # ---- the reference's surface ------------------------------------------------------------------------------------
    def set_lr_scheduler(self):
        return torch.optim.lr_scheduler.LambdaLR(self.optimizer, lr_lambda=lambda epoch: 0.96 ** (epoch / 50))

    def test(self, users_to_test, is_val):
        self.model.eval()
        with torch.no_grad():
            outs = self.model(self.norm_adj, build_item_graph=True)
        return self._evaluate(outs[0], outs[1], users_to_test, is_val)

    def bpr_loss(self, users, pos_items, neg_items):
        """The reference's signature: three gathered [B, d] blocks -> (mf_loss, emb_loss, reg_loss)."""
        mf, emb = ops.bpr(users, pos_items, neg_items, self.decay, self.batch_size)
        return mf, emb, 0.0

    def train(self):
        a, dg = self.args, self.data
        micro = self.model_name == "micro"
        stopping_step, best_recall, test_ret = 0, 0, None
        for epoch in range(a.epoch):
            t1 = time()
            loss, mf_loss, emb_loss, reg_loss, contrastive_loss = 0., 0., 0., 0., 0.
            n_batch = dg.n_train // a.batch_size + 1
            build_item_graph = True
            for idx in range(n_batch):
                self.model.train()
                self.optimizer.zero_grad()
                users, pos_items, neg_items = dg.sample()
                outs = self.model(self.norm_adj, build_item_graph=build_item_graph)
                build_item_graph = False
                batch_mf, batch_emb, batch_reg, batch_cl = self._batch_losses(outs, users, pos_items, neg_items)
                batch_loss = batch_mf + batch_emb + batch_reg
                if micro:
                    batch_loss = batch_loss + batch_cl
                batch_loss.backward()
                self.optimizer.step()
                loss += float(batch_loss.detach())
                mf_loss += float(batch_mf.detach())
                emb_loss += float(batch_emb.detach())
                reg_loss += float(batch_reg)
                if micro:
                    contrastive_loss += float(batch_cl.detach())
            self.lr_scheduler.step()
            if math.isnan(loss):
                raise FloatingPointError("loss is nan")
            evaluate = ((epoch + 1) % a.verbose == 0) if micro else (epoch % a.verbose == 0)
            print("Epoch %d [%.1fs]: train==[%.5f=%.5f + %.5f%s]" % (
                epoch, time() - t1, loss, mf_loss, emb_loss, (" + %.5f" % contrastive_loss) if micro else ""))
            if not evaluate:
                continue
            t2 = time()
            users_to_test = list(dg.test_set.keys())
            users_to_val = list(dg.val_set.keys())
            ret = self.test(users_to_val, is_val=True)
            rec = {"epoch": epoch, "loss": loss, "mf_loss": mf_loss, "emb_loss": emb_loss, "contrastive_loss": contrastive_loss,
                   "val": ret, "test": None}
            if a.verbose > 0:
                print("Epoch %d [%.1fs + %.1fs]:  val==[%.5f=%.5f + %.5f + %.5f], recall=[%.5f, %.5f], precision=[%.5f, %.5f], "
                      "hit=[%.5f, %.5f], ndcg=[%.5f, %.5f]" % (
                          epoch, t2 - t1, time() - t2, loss, mf_loss, emb_loss, reg_loss, ret["recall"][0], ret["recall"][-1],
                          ret["precision"][0], ret["precision"][-1], ret["hit_ratio"][0], ret["hit_ratio"][-1],
                          ret["ndcg"][0], ret["ndcg"][-1]))
            if ret["recall"][1] > best_recall:
                best_recall = ret["recall"][1]
                test_ret = self.test(users_to_test, is_val=False)
                rec["test"] = test_ret
                print("Epoch %d: test== recall=[%.5f, %.5f], precision=[%.5f, %.5f], hit=[%.5f, %.5f], ndcg=[%.5f, %.5f]" % (
                    epoch, test_ret["recall"][0], test_ret["recall"][-1], test_ret["precision"][0], test_ret["precision"][-1],
                    test_ret["hit_ratio"][0], test_ret["hit_ratio"][-1], test_ret["ndcg"][0], test_ret["ndcg"][-1]))
                stopping_step = 0
                self.history.append(rec)
            elif stopping_step < a.early_stopping_patience:
                stopping_step += 1
                self.history.append(rec)
                print("#####Early stopping steps: %d #####" % stopping_step)
            else:
                self.history.append(rec)
                print("#####Early stop! #####")
                break
        print(test_ret)
        return test_ret


def main(model_name="lattice", argv=None):
    """python -m mmssl_amd.baselines_main [lattice|micro] --dataset ... (the reference's `__main__` blocks)."""
    args = parse_args(model_name, argv)
    set_seed(args.seed)
    data = batch_test.init_data(path=args.data_path + args.dataset, batch_size=args.batch_size)
    cfg = {"n_users": data.n_users, "n_items": data.n_items}
    _, norm_adj, _ = data.get_adj_mat()
    cfg["norm_adj"] = norm_adj
    return Trainer(data_config=cfg, args=args, data=data).train()


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 1 and sys.argv[1].lower() in ("lattice", "micro"):
        main(sys.argv[1], sys.argv[2:])
    else:
        main("lattice", sys.argv[1:])
