"""Evaluation: `test_torch(ua_embeddings, ia_embeddings, users_to_test, is_val)` with the
reference's protocol and result dict (/root/reference/MMSSL/utility/batch_test.py:83-169):
score = U_b . I^T in fp32 on the device, drop each user's training items, take the top max(Ks)
by score with ties broken towards the LOWER item id (heapq.nlargest over an ascending-id dict is
a stable sort), then precision / recall / ndcg / hit_ratio @ Ks averaged over the tested users.

The reference ranks with a multiprocessing.Pool + heapq per user on the host (SURVEY.md section 8f
"next #2"); here scoring + masking (one fused fp32-MFMA tile kernel), the top-K selection (one kernel, the heapq
tie rule) and the hit test run on the device of the embeddings through libmmssl_hip.so: no library GEMM, no sort
of the [users, items] matrix, no process pool, no [users, items] device-to-host copy.
"""
import numpy as np
import torch

from .. import config, ops
from . import metrics
from .load_data import Data

data_generator = None     # set by init_data(); the reference builds it at import time


def init_data(path=None, batch_size=None):
    """Create the module-level `data_generator` (reference: batch_test.py:16)."""
    global data_generator
    a = config.args
    data_generator = Data(path=path or (a.data_path + a.dataset), batch_size=batch_size or a.batch_size)
    return data_generator


def get_performance(user_pos_test, r, auc, Ks):
    precision, recall, ndcg, hit_ratio = [], [], [], []
    for K in Ks:
        precision.append(metrics.precision_at_k(r, K))
        recall.append(metrics.recall_at_k(r, K, len(user_pos_test)))
        ndcg.append(metrics.ndcg_at_k(r, K))
        hit_ratio.append(metrics.hit_at_k(r, K))
    return {"recall": np.array(recall), "precision": np.array(precision), "ndcg": np.array(ndcg),
            "hit_ratio": np.array(hit_ratio), "auc": auc}


def _set_csr(data, name, mapping):
    """CSR (indptr, indices) over user ids of a {user: [items]} mapping, cached on the Data object."""
    cache = data.__dict__.setdefault("_eval_csr", {})
    hit = cache.get(name)
    if hit is None or hit[0] is not mapping:
        import itertools
        n_users, n = data.n_users, len(mapping)
        users = np.fromiter((int(u) for u in mapping.keys()), dtype=np.int64, count=n)
        lens = np.fromiter((len(v) for v in mapping.values()), dtype=np.int64, count=n)
        counts = np.zeros(n_users + 1, dtype=np.int64)
        counts[users + 1] = lens
        indptr = np.cumsum(counts)
        total = int(indptr[-1])
        flat = np.fromiter(itertools.chain.from_iterable(mapping.values()), dtype=np.int64, count=total)
        # entry j of `flat` belongs to the k-th user of the mapping's order: its slot = that user's row start + its offset
        first = np.cumsum(lens) - lens
        dest = np.repeat(indptr[users] - first, lens) + np.arange(total, dtype=np.int64)
        indices = np.empty(total, dtype=np.int64)
        indices[dest] = flat
        hit = (mapping, indptr, indices)
        cache[name] = hit
    return hit[1], hit[2]


def _rows_of(indptr, indices, users):
    """(row-in-batch, item) pairs of the given users, vectorised."""
    users = np.asarray(users, dtype=np.int64)
    beg, end = indptr[users], indptr[users + 1]
    cnt = end - beg
    rows = np.repeat(np.arange(len(users), dtype=np.int64), cnt)
    # offsets inside each user's slice
    start_of = np.repeat(beg - np.concatenate(([0], np.cumsum(cnt)[:-1])), cnt)
    cols = indices[np.arange(int(cnt.sum()), dtype=np.int64) + start_of]
    return rows, cols, cnt


def _device_csr(data, name, mapping, device):
    """(rowptr int32, sorted cols int32) of a {user: [items]} mapping on `device`, cached on the Data object."""
    cache = data.__dict__.setdefault("_eval_csr_dev", {})
    key = (name, str(device))
    hit = cache.get(key)
    if hit is None or hit[0] is not mapping:
        indptr, indices = _set_csr(data, name, mapping)
        # per-row ascending (the kernels bisect the rows): one lexsort by (row, column) instead of a Python loop over
        # every user (35 ms of the first test() call on the Baby shape)
        rows = np.repeat(np.arange(len(indptr) - 1, dtype=np.int64), np.diff(indptr))
        cols = indices[np.lexsort((indices, rows))]
        hit = (mapping, torch.as_tensor(indptr, dtype=torch.int32, device=device),
               torch.as_tensor(cols, dtype=torch.int32, device=device))
        cache[key] = hit
    return hit[1], hit[2]


def _metric_sums(hits, n_pos, Ks):
    """Sums over users of precision / recall / ndcg / hit_ratio @ Ks from the binary hit matrix
    [users, max(Ks)] (float64, the formulas of utility/metrics.py applied to all rows at once).
    ndcg's ideal DCG comes from the row's own hit count, like metrics.ndcg_at_k (sorted(r))."""
    hits = np.asarray(hits, dtype=np.float64)
    n_pos = np.asarray(n_pos, dtype=np.float64)
    disc = 1.0 / np.log2(np.arange(2, hits.shape[1] + 2, dtype=np.float64))
    cum_disc = np.concatenate(([0.0], np.cumsum(disc)))
    tot = hits.sum(1).astype(np.int64)                       # hits within max(Ks): the "ideal" list's ones
    out = {k: np.zeros(len(Ks)) for k in ("precision", "recall", "ndcg", "hit_ratio")}
    for j, K in enumerate(Ks):
        h = hits[:, :K]
        s = h.sum(1)
        out["precision"][j] = (s / K).sum()
        out["recall"][j] = np.where(n_pos > 0, s / np.maximum(n_pos, 1.0), 0.0).sum()
        dcg = (h * disc[:K]).sum(1)
        best = cum_disc[np.minimum(tot, K)]
        out["ndcg"][j] = np.where(best > 0, dcg / np.where(best > 0, best, 1.0), 0.0).sum()
        out["hit_ratio"][j] = (s > 0).sum()
    return out


def test_torch(ua_embeddings, ia_embeddings, users_to_test, is_val, drop_flag=False, batch_test_flag=False,
               data=None):
    """Scores, masking and ranking run on the embeddings' device: score = U_b . I^T (fp32), training
    items set to -inf (never candidates), then a STABLE descending sort — equal scores keep ascending
    item id, which is exactly the order heapq.nlargest gives the reference (batch_test.py:21-36).
    Only the [users, max(Ks)] hit matrix travels to the host for the metric formulas."""
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
    dev = ua_embeddings.device
    tr_csr = _device_csr(data, "train", data.train_items, dev)
    pos_name = "val" if is_val else "test"
    po_csr = _device_csr(data, pos_name, pos_of, dev)
    if len(Ks) > 8:
        raise ValueError("test_torch: at most 8 cut-offs in --Ks")
    # every batch: scores with the training items at -inf (batch_test.py:98-100) from ONE fused tile kernel, the top
    # max(Ks) per row by the selection kernel (descending score, ascending id on ties), then the metric formulas on the
    # device, summed in float64 into `acc` - no [users, items] sort, no dense positives matrix, and NOTHING comes back to
    # the host until the last batch is queued (round 5 fetched a hit matrix per batch: 10.3 ms per 1 k users end to end
    # against 0.3 ms of kernels)
    all_users = torch.as_tensor(np.asarray(users_to_test, dtype=np.int64), device=dev)
    acc = torch.zeros((4, 8), dtype=torch.float64, device=dev)
    ws = None
    ua, ia = ua_embeddings.detach(), ia_embeddings.detach()
    count = 0
    for start in range(0, n_test_users, u_batch):
        idx = all_users[start:start + u_batch]
        rate, _ = ops.sim_rows(ua, ia, qidx=idx, mask=tr_csr, mask_value=float("-inf"))
        order = ops.topk_rows(rate, k_max)
        ws = ops.eval_accumulate(po_csr[0], po_csr[1], idx, order, Ks, acc, ws)
        count += int(idx.shape[0])
    assert count == n_test_users
    if n_test_users:
        tot = acc.cpu().numpy()
        for m, key in enumerate(("precision", "recall", "ndcg", "hit_ratio")):
            result[key] = tot[m, :len(Ks)] / n_test_users
    return result
