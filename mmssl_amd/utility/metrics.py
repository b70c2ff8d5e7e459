"""Ranking metrics with the reference's definitions (/root/reference/MMSSL/utility/metrics.py).
`r` is a binary hit list in rank order. Note ndcg's ideal DCG is taken from the sorted hit
list itself (metrics.py:67-71), not from the number of ground-truth items."""
import numpy as np


def precision_at_k(r, k):
    assert k >= 1
    return float(np.mean(np.asarray(r)[:k]))


def recall_at_k(r, k, all_pos_num):
    if all_pos_num == 0:
        return 0
    return float(np.sum(np.asarray(r, dtype=np.float64)[:k]) / all_pos_num)


def dcg_at_k(r, k, method=1):
    r = np.asarray(r, dtype=np.float64)[:k]
    if not r.size:
        return 0.0
    if method == 0:
        return float(r[0] + np.sum(r[1:] / np.log2(np.arange(2, r.size + 1))))
    if method == 1:
        return float(np.sum(r / np.log2(np.arange(2, r.size + 2))))
    raise ValueError("method must be 0 or 1.")


def ndcg_at_k(r, k, method=1):
    best = dcg_at_k(sorted(r, reverse=True), k, method)
    return dcg_at_k(r, k, method) / best if best else 0.0


def hit_at_k(r, k):
    return 1.0 if np.sum(np.asarray(r)[:k]) > 0 else 0.0


def F1(pre, rec):
    return (2.0 * pre * rec) / (pre + rec) if pre + rec > 0 else 0.0


def auc(ground_truth, prediction):
    try:
        from sklearn.metrics import roc_auc_score
        return roc_auc_score(y_true=ground_truth, y_score=prediction)
    except Exception:
        return 0.0
