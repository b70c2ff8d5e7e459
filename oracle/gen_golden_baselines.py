"""Golden vectors for the LATTICE / MICRO baselines (SURVEY.md 8f "next #4") by RUNNING THE UPSTREAM REFERENCE on
CPU - TEST INFRASTRUCTURE, run in the build container only:   python oracle/gen_golden_baselines.py
Each model is imported in its own child process (LATTICE/codes and MICRO/codes both ship a top-level `utility`
package and a `Models` module). The committed .npz files are DATA: inputs + the reference's outputs.
  G10  LATTICE.forward(adj, build_item_graph=True)  (cf_model lightgcn)   LATTICE/codes/Models.py:32-136
  G11  MICRO.forward + batched_contrastive_loss     (cf_model lightgcn)   MICRO/codes/Models.py:13-160
Shims: argv before import (parse_args at import), .cuda() = identity, dense item graph for MICRO (--sparse 0: its
sparse path needs torch_scatter, which this image lacks; the graph is the same)."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
U, I, E, DV, DT, D, K = 160, 96, 1000, 32, 48, 64, 10


def child(which):
    import scipy.sparse as sp
    import torch
    root = "/root/reference/%s/codes" % which
    sys.path.insert(0, root)
    tmp = "/tmp/mmssl_golden_bl_%s/" % which
    os.makedirs(tmp, exist_ok=True)
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    sys.argv = ["main.py", "--data_path", tmp, "--cf_model", "lightgcn", "--topk", str(K)] + (["--sparse", "0"] if which == "MICRO" else [])
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    import Models
    rng = np.random.default_rng(3)
    img = rng.standard_normal((I, DV)).astype(np.float32)
    txt = rng.standard_normal((I, DT)).astype(np.float32)
    # LightGCN adjacency of a random bipartite graph: D^-1/2 (A) D^-1/2 over (U+I) nodes (load_data.py:105-150 'norm_adj'
    # variants are the caller's business; any normalised (U+I)^2 matrix pins the propagation)
    r = rng.integers(0, U, E)
    c = rng.integers(0, I, E)
    R = sp.csr_matrix((np.ones(E, np.float32), (r, c)), shape=(U, I))
    R.data[:] = 1.0
    A = sp.bmat([[None, R], [R.T, None]]).tocsr()
    deg = np.asarray(A.sum(1)).ravel()
    dinv = np.power(deg, -0.5, where=deg > 0, out=np.zeros_like(deg))
    A = (sp.diags(dinv) @ A @ sp.diags(dinv)).tocoo().astype(np.float32)
    adj = torch.sparse_coo_tensor(torch.from_numpy(np.vstack((A.row, A.col)).astype(np.int64)), torch.from_numpy(A.data), A.shape)
    torch.manual_seed(11)
    cls = getattr(Models, which)
    model = cls(U, I, D, [64, 64], [0.1, 0.1], img, txt)
    model.train()
    P = {"p." + k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    outs = model(adj, build_item_graph=True)
    rec = {"adj_row": A.row.astype(np.int64), "adj_col": A.col.astype(np.int64), "adj_val": A.data, "n_users": U, "n_items": I,
           "topk": K, "image_feat": img, "text_feat": txt, **P}
    names = ["ua", "ia"] if which == "LATTICE" else ["ua", "ia", "image_item", "text_item", "h"]
    for n, o in zip(names, outs):
        rec["o." + n] = o.detach().numpy()
    # a fixed scalar of the outputs -> gradients of the parameters that shape the learned item graph
    i_ = np.arange(U + I, dtype=np.float64)[:, None]
    j_ = np.arange(D, dtype=np.float64)[None, :]
    cot = torch.from_numpy(np.sin(0.37 * i_ + 1.3 * j_).astype(np.float32))
    scalar = (outs[0] * cot[:U]).sum() + (outs[1] * cot[U:]).sum()
    if which == "MICRO":
        cl = model.batched_contrastive_loss(outs[2], outs[4]) + model.batched_contrastive_loss(outs[3], outs[4])
        rec["cl"] = np.float32(cl.item())
        scalar = scalar + 0.03 * cl
    scalar.backward()
    rec["scalar"] = np.float32(scalar.item())
    for k, p in model.named_parameters():
        if p.grad is not None and k.split(".")[0] in ("image_trs", "text_trs", "modal_weight", "item_id_embedding",
                                                      "user_embedding", "query"):
            rec["g." + k] = p.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "g1%d_%s_lightgcn.npz" % (0 if which == "LATTICE" else 1, which.lower())), **rec)
    print(which, "scalar", float(scalar), {k: v.shape for k, v in rec.items() if k.startswith("o.")})


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for w in ("LATTICE", "MICRO"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), w], capture_output=True, text=True)
            print(r.stdout[-600:], r.stderr[-1500:] if r.returncode else "")
