"""Seeded synthetic MMSSL dataset writer (TEST INFRASTRUCTURE — oracle side).

Writes the on-disk layout the reference loads (reference MMSSL/main.py:54-58 and
MMSSL/utility/load_data.py:14-27): {train,val,test}.json as {"uid": [item, ...]},
image_feat.npy, text_feat.npy, and a pickled scipy CSR `train_mat` (U x I, float32).

Only tests/, bench.py's cpu_baseline leg and oracle/gen_golden.py use this file.
"""
import json
import os
import pickle

import numpy as np
import scipy.sparse as sp


def make_interactions(n_users, n_items, n_edges, seed=1, min_deg=3, item_alpha=0.8):
    """Bipartite interaction lists with Zipf-like user degrees (>= min_deg) and item
    popularity ~ rank^-alpha, no duplicate edges."""
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, n_users + 1) ** 0.6
    rng.shuffle(w)
    deg = np.maximum(min_deg, np.floor(w / w.sum() * n_edges)).astype(np.int64)
    deg = np.minimum(deg, n_items // 2)
    p = 1.0 / np.arange(1, n_items + 1) ** item_alpha
    p = p[rng.permutation(n_items)]
    p /= p.sum()
    rows = []
    for u in range(n_users):
        k = int(deg[u])
        items = np.unique(rng.choice(n_items, size=k * 2 + 4, p=p))
        rng.shuffle(items)
        rows.append(np.sort(items[:k]))
    return rows


def write_dataset(root, name, n_users, n_items, n_edges, dv, dt, seed=1, holdout=1):
    """Create `<root>/<name>/` in the reference's format. Returns the directory."""
    d = os.path.join(root, name)
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(seed + 7)
    rows = make_interactions(n_users, n_items, n_edges, seed=seed)
    train, val, test = {}, {}, {}
    for u, items in enumerate(rows):
        items = [int(i) for i in items]
        rng.shuffle(items)
        if len(items) >= 3 + 2 * holdout:
            test[str(u)] = sorted(items[:holdout])
            val[str(u)] = sorted(items[holdout:2 * holdout])
            train[str(u)] = sorted(items[2 * holdout:])
        else:
            train[str(u)] = sorted(items)
            val[str(u)] = []
            test[str(u)] = []
    # make sure the largest item / user id appears so n_users/n_items are as requested
    train[str(n_users - 1)] = sorted(set(train[str(n_users - 1)]) | {n_items - 1})
    for nm, obj in (("train", train), ("val", val), ("test", test)):
        with open(os.path.join(d, nm + ".json"), "w") as f:
            json.dump(obj, f)
    r, c = [], []
    for u, items in train.items():
        r += [int(u)] * len(items)
        c += items
    mat = sp.csr_matrix((np.ones(len(r), np.float32), (r, c)), shape=(n_users, n_items))
    with open(os.path.join(d, "train_mat"), "wb") as f:
        pickle.dump(mat, f)
    np.save(os.path.join(d, "image_feat.npy"), rng.standard_normal((n_items, dv)).astype(np.float32))
    np.save(os.path.join(d, "text_feat.npy"), rng.standard_normal((n_items, dt)).astype(np.float32))
    return d
