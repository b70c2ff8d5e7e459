"""Seeded synthetic workloads of the BASELINE.json shapes (no datasets ship with the reference;
its README only gives sizes: /root/reference/README.md:40-47).

Bipartite user-item interaction graph with Zipf-like user degrees (min 3) and item popularity
~ rank^-0.8, no duplicate edges; fully vectorised so the Amazon-Baby shape
takes well under a second and shards of the 100M-edge stress shape are feasible.
"""
import numpy as np
import scipy.sparse as sp

SHAPES = {
    # name: (n_users, n_items, n_edges, dv, dt)
    "tiktok": (9319, 6710, 59541, 128, 768),
    "baby": (35598, 18357, 256308, 4096, 1024),
    "sports": (19445, 7050, 139110, 4096, 1024),
    "allrecipes": (19805, 10067, 58922, 2048, 20),
    "synth": (2_000_000, 1_000_000, 100_000_000, 0, 0),
}


def interaction_matrix(n_users, n_items, n_edges, seed=1, min_deg=3, user_alpha=0.6, item_alpha=0.8):
    """scipy CSR [n_users, n_items] of ones with exactly min(n_edges, feasible) unique edges."""
    rng = np.random.default_rng(seed)
    w = 1.0 / np.arange(1, n_users + 1, dtype=np.float64) ** user_alpha
    rng.shuffle(w)
    deg = np.maximum(min_deg, np.floor(w / w.sum() * n_edges)).astype(np.int64)
    deg = np.minimum(deg, max(1, n_items // 2))
    p = 1.0 / np.arange(1, n_items + 1, dtype=np.float64) ** item_alpha
    p = p[rng.permutation(n_items)]
    cdf = np.cumsum(p / p.sum())
    draw = (deg * 1.35 + 4).astype(np.int64)
    users = np.repeat(np.arange(n_users, dtype=np.int64), draw)
    items = np.searchsorted(cdf, rng.random(users.shape[0]), side="right").astype(np.int64)
    items = np.minimum(items, n_items - 1)
    key = np.unique(users * n_items + items)
    if key.shape[0] > n_edges:
        key = np.sort(rng.choice(key, size=n_edges, replace=False))
    users, items = key // n_items, key % n_items
    return sp.csr_matrix((np.ones(key.shape[0], np.float32), (users, items)), shape=(n_users, n_items))


def normalised_pair(raw):
    """(A_ui, A_iu) scipy CSR, each scaled by 1/sqrt(own row degree) — what Trainer.csr_norm
    (mean_flag=True) yields for a 0/1 matrix (reference MMSSL/main.py:65-67, 89-103)."""
    def norm(m):
        m = m.tocsr().astype(np.float32)
        d = np.asarray(m.sum(1)).ravel()
        s = np.power(d + 1e-8, -0.5).astype(np.float64)
        return (sp.diags(s) @ m).tocsr().astype(np.float32)
    return norm(raw), norm(raw.T.tocsr())


def spmm_bytes(csr, d):
    """Algorithmic bytes of one SpMM launch (gather-per-edge, int32 indices):
    nnz*(4 col + 4 val + 4d) + rows*4d + (rows+1)*4."""
    return csr.nnz * (8 + 4 * d) + csr.shape[0] * 4 * d + (csr.shape[0] + 1) * 4
