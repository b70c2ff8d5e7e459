"""Golden vectors for the LATTICE / MICRO baselines (SURVEY.md 8f "next #4") by RUNNING THE UPSTREAM REFERENCE on
CPU - TEST INFRASTRUCTURE, run in the build container only:   python oracle/gen_golden_baselines.py
Each model is imported in its own child process (LATTICE/codes and MICRO/codes both ship a top-level `utility`
package and a `Models` module). The committed .npz files are DATA: inputs + the reference's outputs.
  G10  LATTICE.forward(adj, build_item_graph=True)  (cf_model lightgcn)   LATTICE/codes/Models.py:32-136
  G11  MICRO.forward + batched_contrastive_loss     (cf_model lightgcn)   MICRO/codes/Models.py:13-160
  G13  LATTICE (cf_model ngcf) and the NGCF class                          LATTICE/codes/Models.py:106-118, MICRO/codes/Models.py:179-217
  G14  MICRO (cf_model ngcf) with --sparse 1                               MICRO/codes/Models.py:126-139, utility/norm.py:8-36
Shims: argv before import (parse_args at import), .cuda() = identity. G10 / G11 use the dense item graph (--sparse 0).
MICRO's sparse path imports torch_scatter, which this image lacks: for G14 THIS GENERATOR (and nothing else) installs a
stand-in module whose scatter_add is index_add_ - the one function utility/norm.py takes from it - and also checks that
the sparse path's lightgcn outputs equal G11's (dense path) to 1e-6: the two paths build the same graph.
NGCF goldens use dropout rates 0.0 (nn.Dropout is then the identity in training mode: no random stream to record)."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden")
U, I, E, DV, DT, D, K = 160, 96, 1000, 32, 48, 64, 10


def child(which, cf_model="lightgcn", sparse=0, cls_name=None, tag=None):
    import scipy.sparse as sp
    import torch
    root = "/root/reference/%s/codes" % which
    sys.path.insert(0, root)
    tmp = "/tmp/mmssl_golden_bl_%s/" % which
    os.makedirs(tmp, exist_ok=True)
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    sys.argv = ["main.py", "--data_path", tmp, "--cf_model", cf_model, "--topk", str(K)] + (
        ["--sparse", str(sparse)] if which == "MICRO" else [])
    if sparse:
        import types
        ts = types.ModuleType("torch_scatter")

        def scatter_add(src, index, dim=0, dim_size=None):
            assert dim == 0
            return torch.zeros(dim_size, dtype=src.dtype).index_add_(0, index, src)
        ts.scatter_add = scatter_add
        sys.modules["torch_scatter"] = ts
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
    cls_name = cls_name or which
    cls = getattr(Models, cls_name)
    drops = [0.0, 0.0] if cf_model == "ngcf" or cls_name == "NGCF" else [0.1, 0.1]
    model = cls(U, I, D, [64, 64], drops, img, txt)
    model.train()
    P = {"p." + k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    outs = model(adj, build_item_graph=True)
    rec = {"adj_row": A.row.astype(np.int64), "adj_col": A.col.astype(np.int64), "adj_val": A.data, "n_users": U, "n_items": I,
           "topk": K, "image_feat": img, "text_feat": txt, **P}
    names = ["ua", "ia"] if cls_name in ("LATTICE", "NGCF") else ["ua", "ia", "image_item", "text_item", "h"]
    for n, o in zip(names, outs):
        rec["o." + n] = o.detach().numpy()
    # a fixed scalar of the outputs -> gradients of the parameters that shape the learned item graph
    width = outs[0].shape[1]
    i_ = np.arange(U + I, dtype=np.float64)[:, None]
    j_ = np.arange(width, dtype=np.float64)[None, :]
    cot = torch.from_numpy(np.sin(0.37 * i_ + 1.3 * j_).astype(np.float32))
    scalar = (outs[0] * cot[:U]).sum() + (outs[1] * cot[U:]).sum()
    if cls_name == "MICRO":
        cl = model.batched_contrastive_loss(outs[2], outs[4]) + model.batched_contrastive_loss(outs[3], outs[4])
        rec["cl"] = np.float32(cl.item())
        scalar = scalar + 0.03 * cl
    scalar.backward()
    rec["scalar"] = np.float32(scalar.item())
    for k, p in model.named_parameters():
        if p.grad is not None and k.split(".")[0] in ("image_trs", "text_trs", "modal_weight", "item_id_embedding",
                                                      "user_embedding", "query", "GC_Linear_list", "Bi_Linear_list"):
            g = p.grad
            rec["g." + k] = (g.to_dense() if g.is_sparse else g).numpy()
    if tag is None:
        tag = "g1%d_%s_lightgcn" % (0 if which == "LATTICE" else 1, which.lower())
    if sparse and cf_model == "lightgcn":            # the sparse path builds the same graph as the dense one (G11)
        ref = np.load(os.path.join(OUT, "g11_micro_lightgcn.npz"))
        for n in names:
            err = float(np.abs(rec["o." + n] - ref["o." + n]).max() / (np.abs(ref["o." + n]).max() + 1e-30))
            assert err < 1e-6, (n, err)
        print("MICRO --sparse 1 == --sparse 0 (G11) on every output")
        return
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **rec)
    print(which, cls_name, cf_model, "sparse", sparse, "scalar", float(scalar),
          {k: v.shape for k, v in rec.items() if k.startswith("o.")}, sorted(k for k in rec if k.startswith("g.")))


CASES = {"LATTICE": ("LATTICE", "lightgcn", 0, None, None),
         "MICRO": ("MICRO", "lightgcn", 0, None, None),
         "LATTICE_ngcf": ("LATTICE", "ngcf", 0, None, "g13_lattice_ngcf"),
         "NGCF": ("MICRO", "ngcf", 0, "NGCF", "g13_ngcf"),
         "MICRO_sparse_check": ("MICRO", "lightgcn", 1, None, None),
         "MICRO_ngcf_sparse": ("MICRO", "ngcf", 1, None, "g14_micro_ngcf_sparse")}


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(*CASES[sys.argv[1]])
    else:
        for w in CASES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), w], capture_output=True, text=True)
            print(r.stdout[-600:], r.stderr[-1500:] if r.returncode else "")
