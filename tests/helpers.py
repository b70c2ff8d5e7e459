"""Shared test helpers: load golden fixtures, rebuild graphs / datasets from them."""
import json
import os
import pickle

import numpy as np
import scipy.sparse as sp
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

OUT_NAMES = ["ua", "ia", "image_item", "text_item", "image_user", "text_user",
             "ua2", "ia2", "image_user_id", "text_user_id", "image_item_id", "text_item_id"]


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def dataset():
    d = load("dataset_tiny.npz")
    U, I = int(d["n_users"]), int(d["n_items"])
    raw = sp.csr_matrix((d["tm_val"], (d["tm_row"], d["tm_col"])), shape=(U, I))
    return d, raw, U, I


def split_lists(d, nm):
    out, k = {}, 0
    for uid, ln in zip(d[nm + "_uid"], d[nm + "_len"]):
        out[int(uid)] = [int(x) for x in d[nm + "_items"][k:k + int(ln)]]
        k += int(ln)
    return out


def write_dataset_dir(root, name="tiny"):
    """Re-create the on-disk dataset (reference layout) from the fixture. Returns parent dir
    WITH trailing slash (the reference concatenates data_path + dataset)."""
    d, raw, U, I = dataset()
    dd = os.path.join(root, name)
    os.makedirs(dd, exist_ok=True)
    for nm in ("train", "val", "test"):
        lists = split_lists(d, nm)
        with open(os.path.join(dd, nm + ".json"), "w") as f:
            json.dump({str(k): v for k, v in lists.items()}, f)
    np.save(os.path.join(dd, "image_feat.npy"), d["image_feat"])
    np.save(os.path.join(dd, "text_feat.npy"), d["text_feat"])
    with open(os.path.join(dd, "train_mat"), "wb") as f:
        pickle.dump(raw, f)
    return os.path.join(root, "")


def modal_raw(fx, which, U, I):
    r, c, v = fx["modal_%s_row" % which], fx["modal_%s_col" % which], fx["modal_%s_val" % which]
    return sp.csr_matrix((v, (r, c)), shape=(U, I))


def params(fx, device="cpu", requires_grad=False):
    P = {}
    for k in fx.files:
        if k.startswith("p."):
            t = torch.from_numpy(fx[k]).to(device)
            if requires_grad:
                t.requires_grad_(True)
            P[k[2:]] = t
    return P


def cotangent(k, shape):
    """Must match oracle/gen_golden.py:cotangent."""
    i = np.arange(shape[0], dtype=np.float64)[:, None]
    j = np.arange(shape[1], dtype=np.float64)[None, :]
    return np.sin(0.37 * i + 1.3 * j + 0.71 * k).astype(np.float32)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def row_rel(a, b, floor=1e-3):
    """max over rows of |a_r - b_r|_inf / max(|b_r|_inf, floor * |b|_inf): a row with a small gradient is checked against
    its OWN scale (down to `floor` of the table's largest entry), not against the largest row of the table."""
    a = torch.as_tensor(np.asarray(a.detach().cpu() if torch.is_tensor(a) else a)).double()
    b = torch.as_tensor(np.asarray(b.detach().cpu() if torch.is_tensor(b) else b)).double()
    if a.dim() == 1:
        a, b = a[None], b[None]
    den = torch.clamp(b.abs().amax(1), min=floor * float(b.abs().max()) + 1e-30)
    return float(((a - b).abs().amax(1) / den).max())


def check_grad(got, want, tol, what, row_tol=5e-3, floor=1e-3):
    """A gradient against its reference: max-abs / max-abs over the whole tensor below `tol` AND - for the embedding-table
    shaped ones ([rows, d] with many rows: user / item tables and their per-row relatives) - every row within `row_tol`
    of its own scale."""
    e = rel_err(got.detach().cpu() if torch.is_tensor(got) else got, want.detach().cpu() if torch.is_tensor(want) else want)
    assert e < tol, (what, e)
    w = np.asarray(want.detach().cpu() if torch.is_tensor(want) else want)
    if w.ndim == 2 and w.shape[0] >= 64 and w.shape[0] > 4 * w.shape[1]:
        er = row_rel(got, want, floor)
        assert er < row_tol, (what, "row-wise", er)
    return e
