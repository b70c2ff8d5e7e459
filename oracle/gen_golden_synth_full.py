"""Golden vectors of BASELINE configs[4] AT ITS REAL SIZE (2M users x 1M items, 100M edges, d = 128, 3-layer GCN): the
CPU oracle's generator step on the seeded problem of mmssl_amd.synth (stress_blocks / stress_inputs) -> the loss terms, the
gradients of every small parameter, and SAMPLED ROWS of the two output tables and of the two embedding-table gradients
(tests/golden/synth_full_n1.npz, ~2 MB). Run in the build container (CPU, ~25 GB of RAM, minutes):

    python oracle/gen_golden_synth_full.py [--scale 8] [--out tests/golden/synth_full_n1.npz]

Re-running it reproduces the committed file bit for bit (checked in the build container, 8 threads: ~4 min, ~30 GB).

TEST INFRASTRUCTURE. The GPU tests (tests/test_synth_full_gpu.py) rebuild the same inputs from the same seeds on the GPU
box and compare the HIP step - one GPU, and 8 ranks sharing it - against these rows; nothing of the oracle runs there.
The reference itself cannot run this size at all (it allocates dense U x I matrices, /root/reference/MMSSL/main.py:59-60),
so the pinned oracle (tests/test_oracle_golden.py) is the only possible witness.

The oracle's functions are used unchanged except for two measures this size forces:
  * memory: modality_attention materialises a [heads, 2, 2, rows, d] tensor (16 GB for 2M users), so it is applied to row
    blocks of 2^16 rows under torch.utils.checkpoint - it is row-independent, the values are those of the whole-table call;
  * the SpMM sums are ACCUMULATED IN FLOAT64 and rounded once to fp32 (forward and the autograd transpose product). A hub
    item of this graph has ~10^6 edges; torch's fp32 COO product adds them one after the other, and a 10^6-term fp32
    running sum is itself only good to ~1e-3 - 1e-4 (first attempt: fp32 oracle vs HIP 4e-4 on the output rows, the HIP
    kernel's 128-edge partial sums being the MORE accurate side). With exactly rounded sums the golden measures the HIP
    path's own error instead of the witness's. Everything else (projection, softmax, normalise, losses) stays fp32.
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

import mmssl_oracle as O      # noqa: E402
from mmssl_amd import synth   # noqa: E402


def sample_rows(batch, n_users, n_items, n=1024):
    """Row ids whose outputs / gradients are kept: half from the batch (the rows the losses read), half anywhere."""
    rng = np.random.default_rng(5)
    u = np.unique(np.concatenate([batch[0].numpy()[: n // 2], rng.integers(0, n_users, n // 2)]))
    i = np.unique(np.concatenate([batch[1].numpy()[: n // 4], batch[2].numpy()[: n // 4], rng.integers(0, n_items, n // 2)]))
    return u.astype(np.int64), i.astype(np.int64)


class _Graph64:
    """A sparse matrix held as float64 CSR together with its transpose (the autograd product)."""

    def __init__(self, mat):
        def csr64(m):
            m = sp.csr_matrix(m).astype(np.float64)
            m.sort_indices()
            return torch.sparse_csr_tensor(torch.from_numpy(m.indptr.astype(np.int64)), torch.from_numpy(m.indices.astype(np.int64)),
                                           torch.from_numpy(m.data), size=m.shape)
        self.A, self.AT = csr64(mat), csr64(mat.T.tocsr())


def _mm64(A, X, cols=32):
    out = torch.empty((A.shape[0], X.shape[1]), dtype=torch.float32)
    for c in range(0, X.shape[1], cols):
        out[:, c:c + cols] = torch.sparse.mm(A, X[:, c:c + cols].double()).float()
    return out


class _Spmm64(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, G):
        ctx.G = G
        return _mm64(G.A, X)

    @staticmethod
    def backward(ctx, g):
        return _mm64(ctx.G.AT, g.contiguous()), None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    out = a.out or os.path.join(ROOT, "tests", "golden", "synth_full_n1.npz" if a.scale == 1 else "synth_full_s%d.npz" % a.scale)
    t0 = time.time()
    raw = sp.vstack(synth.stress_blocks(8, a.scale)).tocsr()
    U, I = raw.shape
    print("graph %d x %d, %d edges  (%.0f s)" % (U, I, raw.nnz, time.time() - t0), flush=True)
    ui, iu = synth.normalised_pair(raw)
    d = 128
    pb = synth.stress_inputs(U, I, d)
    cfg = O.Cfg(embed_size=d, n_ui_layers=3, drop_rate=0.2, batch_size=pb["batch"].shape[1])
    nnz = int(ui.nnz)
    A_ui, A_iu = _Graph64(ui), _Graph64(iu)
    plain = O.spmm
    O.spmm = lambda A, X: _Spmm64.apply(X, A) if isinstance(A, _Graph64) else plain(A, X)
    e_ui = O.to_torch_sparse(sp.csr_matrix((U, I), dtype=np.float32))
    e_iu = O.to_torch_sparse(sp.csr_matrix((I, U), dtype=np.float32))
    del ui, iu, raw
    P = {k: v.clone().requires_grad_(True) for k, v in pb["state"].items()}

    whole = O.modality_attention

    def blocked(Pm, a_, b_, cfg_, rows=1 << 16):
        from torch.utils.checkpoint import checkpoint
        outs = [checkpoint(lambda x, y: whole(Pm, x, y, cfg_), a_[r:r + rows], b_[r:r + rows], use_reentrant=False)
                for r in range(0, a_.shape[0], rows)]
        return torch.cat(outs, 1)
    O.modality_attention = blocked
    users, pos, neg = pb["batch"]
    t0 = time.time()
    o = O.forward(P, pb["img"], pb["txt"], (A_ui, A_iu, e_ui, e_iu, e_ui, e_iu), cfg, training=True,
                  keep_masks=[k.float() for k in pb["keep"]])
    print("forward %.0f s" % (time.time() - t0), flush=True)
    mf, emb, _ = O.bpr(o[0][users], o[1][pos], o[1][neg], cfg.decay, cfg.batch_size)
    feat = O.feat_reg(o[2], o[3], o[4], o[5], I, cfg.feat_reg_decay)
    cl_i, cl_t = O.infonce(o[8][users], o[6][users], cfg.tau), O.infonce(o[9][users], o[6][users], cfg.tau)
    total = mf + emb + feat + cfg.cl_rate * (cl_i + cl_t)
    t0 = time.time()
    total.backward()
    print("backward %.0f s   loss %.7f" % (time.time() - t0, float(total)), flush=True)
    ru, ri = sample_rows(pb["batch"], U, I)
    gu, gi = P["user_id_embedding.weight"].grad, P["item_id_embedding.weight"].grad
    rec = {"scale": np.int64(a.scale), "shape": np.array([U, I, nnz], np.int64),
           "loss": np.array([float(x) for x in (total, mf, emb, feat, cl_i, cl_t)], np.float64),
           "rows_u": ru, "rows_i": ri,
           "ua_rows": o[0].detach()[ru].numpy(), "ia_rows": o[1].detach()[ri].numpy(),
           "g_Eu_rows": gu[ru].numpy(), "g_Ei_rows": gi[ri].numpy(),
           "g_Eu_absmax": np.float32(gu.abs().max()), "g_Ei_absmax": np.float32(gi.abs().max()),
           "g_Eu_sumsq": np.float64((gu.double() ** 2).sum()), "g_Ei_sumsq": np.float64((gi.double() ** 2).sum())}
    for k in ("image_trans.weight", "image_trans.bias", "text_trans.weight", "text_trans.bias",
              "weight_dict.w_self_attention_cat"):
        rec["g_" + k] = P[k].grad.numpy()
    np.savez_compressed(out, **rec)
    print("wrote", out, os.path.getsize(out) >> 10, "KB")


if __name__ == "__main__":
    main()
