"""Evaluation: `test_torch(ua_embeddings, ia_embeddings, users_to_test, is_val)` with the
reference's protocol and result dict (/root/reference/MMSSL/utility/batch_test.py:83-169):
score = U_b . I^T in fp32 on the device, drop each user's training items, take the top max(Ks)
by score with ties broken towards the LOWER item id (heapq.nlargest over an ascending-id dict is
a stable sort), then precision / recall / ndcg / hit_ratio @ Ks averaged over the tested users.

The reference ranks with a multiprocessing.Pool + heapq per user on the host; here ranking is a
vectorised stable argsort per user batch (same order, no process pool). Device top-K is the
SURVEY.md section 8f "next #2" item.
"""
import numpy as np
import torch

from .. import config
from . import metrics
from .load_data import Data

data_generator = None     # set by init_data(); the reference builds it at import time


def init_data(path=None, batch_size=None):
    """Create the module-level `data_generator` (reference: batch_test.py:16)."""
    global data_generator
    a = config.args
    data_generator = Data(path=path or (a.data_path + a.dataset), batch_size=batch_size or a.batch_size)
    return data_generator


def _hit_list(order, pos_mask_row, k_max):
    return pos_mask_row[order[:k_max]].astype(np.int64).tolist()


def get_performance(user_pos_test, r, auc, Ks):
    precision, recall, ndcg, hit_ratio = [], [], [], []
    for K in Ks:
        precision.append(metrics.precision_at_k(r, K))
        recall.append(metrics.recall_at_k(r, K, len(user_pos_test)))
        ndcg.append(metrics.ndcg_at_k(r, K))
        hit_ratio.append(metrics.hit_at_k(r, K))
    return {"recall": np.array(recall), "precision": np.array(precision), "ndcg": np.array(ndcg),
            "hit_ratio": np.array(hit_ratio), "auc": auc}


def test_torch(ua_embeddings, ia_embeddings, users_to_test, is_val, drop_flag=False, batch_test_flag=False,
               data=None):
    data = data or data_generator
    if data is None:
        raise RuntimeError("batch_test.init_data() must be called (or pass data=...)")
    Ks = eval(config.args.Ks) if isinstance(config.args.Ks, str) else list(config.args.Ks)
    k_max = max(Ks)
    n_items = data.n_items
    result = {"precision": np.zeros(len(Ks)), "recall": np.zeros(len(Ks)), "ndcg": np.zeros(len(Ks)),
              "hit_ratio": np.zeros(len(Ks)), "auc": 0.0}
    u_batch = config.args.batch_size * 2
    n_test_users = len(users_to_test)
    pos_of = data.val_set if is_val else data.test_set
    count = 0
    for start in range(0, max(n_test_users, 1), u_batch):
        user_batch = users_to_test[start:start + u_batch]
        if not len(user_batch):
            continue
        idx = torch.as_tensor(user_batch, dtype=torch.int64, device=ua_embeddings.device)
        rate = torch.matmul(ua_embeddings[idx], ia_embeddings.t()).detach().cpu().numpy()
        for row, u in zip(rate, user_batch):
            row = row.copy()
            seen = data.train_items.get(u, [])
            row[np.asarray(seen, dtype=np.int64)] = -np.inf      # never candidates (batch_test.py:98-100)
            order = np.argsort(-row, kind="stable")[:k_max]      # ties -> lower item id first
            pos = pos_of[u]
            posset = set(pos)
            r = [1 if int(i) in posset else 0 for i in order]
            re = get_performance(pos, r, 0.0, Ks)
            for key in ("precision", "recall", "ndcg", "hit_ratio"):
                result[key] += re[key] / n_test_users
            result["auc"] += re["auc"] / n_test_users
            count += 1
    assert count == n_test_users
    return result
