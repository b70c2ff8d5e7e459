"""`Data`: dataset loader + BPR triple sampler with the reference's interface and on-disk format
(/root/reference/MMSSL/utility/load_data.py:10-197): <path>/{train,val,test}.json as
{"uid": [item, ...]}.

`sample()` is BIT-EXACT with the reference: it stays host Python and consumes the global
`random` / `numpy.random` streams in the identical order (users; then per user one positive draw
and rejection-sampled negative draws, interleaved) — SURVEY.md 8a-9. It is deliberately not
vectorised or moved to the GPU.
"""
import json
import random as rd
from time import time

import numpy as np
import scipy.sparse as sp


class Data(object):
    def __init__(self, path, batch_size):
        self.path = path
        self.batch_size = batch_size
        with open(path + "/train.json") as f:
            train = json.load(f)
        with open(path + "/test.json") as f:
            test = json.load(f)
        with open(path + "/val.json") as f:
            val = json.load(f)

        self.neg_pools = {}
        self.exist_users = []
        self.train_items, self.test_set, self.val_set = {}, {}, {}
        max_item, max_user = 0, 0
        self.n_train = self.n_test = self.n_val = 0
        for uid, items in train.items():
            if len(items) == 0:
                continue
            u = int(uid)
            self.exist_users.append(u)
            self.train_items[u] = items
            max_item = max(max_item, max(items))
            max_user = max(max_user, u)
            self.n_train += len(items)
        for split, store, counter in ((test, self.test_set, "n_test"), (val, self.val_set, "n_val")):
            for uid, items in split.items():
                if len(items) == 0:
                    continue
                store[int(uid)] = items
                max_item = max(max_item, max(items))
                setattr(self, counter, getattr(self, counter) + len(items))
        self.n_items = max_item + 1
        self.n_users = max_user + 1
        self.print_statistics()

        rows = np.fromiter((u for u, it in self.train_items.items() for _ in it), dtype=np.int64,
                           count=self.n_train)
        cols = np.fromiter((i for it in self.train_items.values() for i in it), dtype=np.int64,
                           count=self.n_train)
        coo = sp.coo_matrix((np.ones(self.n_train, np.float32), (rows, cols)), shape=(self.n_users, self.n_items))
        self._R_csr = coo.tocsr()
        self._R_csr.data[:] = 1.0
        self._R = None
        self.R_Item_Interacts = sp.dok_matrix((self.n_items, self.n_items), dtype=np.float32)

    @property
    def R(self):
        """User-item interaction matrix as scipy dok (built lazily from the CSR)."""
        if self._R is None:
            self._R = self._R_csr.todok()
        return self._R

    # ---- LightGCN-style (U+I)^2 adjacency used by the LATTICE / MICRO baselines ----------------
    def get_adj_mat(self):
        try:
            t1 = time()
            adj_mat = sp.load_npz(self.path + "/s_adj_mat.npz")
            norm_adj_mat = sp.load_npz(self.path + "/s_norm_adj_mat.npz")
            mean_adj_mat = sp.load_npz(self.path + "/s_mean_adj_mat.npz")
            print("already load adj matrix", adj_mat.shape, time() - t1)
        except Exception:
            adj_mat, norm_adj_mat, mean_adj_mat = self.create_adj_mat()
            sp.save_npz(self.path + "/s_adj_mat.npz", adj_mat)
            sp.save_npz(self.path + "/s_norm_adj_mat.npz", norm_adj_mat)
            sp.save_npz(self.path + "/s_mean_adj_mat.npz", mean_adj_mat)
        return adj_mat, norm_adj_mat, mean_adj_mat

    def create_adj_mat(self):
        R = self._R_csr
        adj = sp.bmat([[None, R], [R.T, None]], format="csr", dtype=np.float32)

        def row_normalised(a):
            deg = np.asarray(a.sum(1)).ravel()
            with np.errstate(divide="ignore"):
                inv = np.power(deg, -1.0)
            inv[np.isinf(inv)] = 0.0
            return sp.diags(inv).dot(a).tocsr()

        return adj, row_normalised(adj + sp.eye(adj.shape[0])), row_normalised(adj)

    # ---- BPR triples -----------------------------------------------------------------------------
    def sample(self):
        if self.batch_size <= self.n_users:
            users = rd.sample(self.exist_users, self.batch_size)
        else:
            users = [rd.choice(self.exist_users) for _ in range(self.batch_size)]
        pos_items, neg_items = [], []
        n_items, randint = self.n_items, np.random.randint
        for u in users:
            mine = self.train_items[u]
            pos_items.append(mine[randint(low=0, high=len(mine), size=1)[0]])
            while True:
                cand = randint(low=0, high=n_items, size=1)[0]
                if cand not in mine:
                    neg_items.append(cand)
                    break
        return users, pos_items, neg_items

    def print_statistics(self):
        print("n_users=%d, n_items=%d" % (self.n_users, self.n_items))
        print("n_interactions=%d" % (self.n_train + self.n_test))
        print("n_train=%d, n_test=%d, sparsity=%.5f" % (
            self.n_train, self.n_test, (self.n_train + self.n_test) / (self.n_users * self.n_items)))
