"""Golden vectors for the LATTICE / MICRO baselines (SURVEY.md 8f "next #4") by RUNNING THE UPSTREAM REFERENCE on
CPU - TEST INFRASTRUCTURE, run in the build container only:   python oracle/gen_golden_baselines.py
Each model is imported in its own child process (LATTICE/codes and MICRO/codes both ship a top-level `utility`
package and a `Models` module). The committed .npz files are DATA: inputs + the reference's outputs.
  G10  LATTICE.forward(adj, build_item_graph=True)  (cf_model lightgcn)   LATTICE/codes/Models.py:32-136
  G11  MICRO.forward + batched_contrastive_loss     (cf_model lightgcn)   MICRO/codes/Models.py:13-160
  G13  LATTICE (cf_model ngcf) and the NGCF class                          LATTICE/codes/Models.py:106-118, MICRO/codes/Models.py:179-217
  G14  MICRO (cf_model ngcf) with --sparse 1                               MICRO/codes/Models.py:126-139, utility/norm.py:8-36
  G15  Trainer.train() of LATTICE and of MICRO: 3 epochs x 7 batches on the tiny dataset of tests/golden/dataset_tiny.npz
       (every sampled batch, per-batch loss terms, parameters before / after, eval-mode embeddings, the validation /
       test metrics of every epoch)                                        LATTICE/codes/main.py:23-185, MICRO/codes/main.py:24-190
       The same child then runs the PRODUCT loop (mmssl_amd/baselines_main.Trainer) over the reference's model class,
       optim.Adam and test_torch on CPU and asserts the identical trajectory: the loop's bookkeeping (sampling order,
       item-graph rebuild flag, scheduler, validation cadence) is pinned here, the HIP models by tests/test_model_gpu.py.
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


TRAINER_ARGV = ["--dataset", "tiny", "--batch_size", "128", "--epoch", "3", "--verbose", "1", "--topk", "10", "--seed", "7",
                "--Ks", "[10, 20]", "--lr", "0.005"]


def _shims(torch, sparse):
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
    torch.cuda.manual_seed_all = lambda seed: None
    import multiprocessing
    if multiprocessing.cpu_count() < 5:          # batch_test.py: Pool(cpu_count() // 5)
        multiprocessing.cpu_count = lambda: 5
    if not hasattr(np, "asfarray"):
        np.asfarray = lambda a, dtype=float: np.asarray(a, dtype=dtype)


def trainer_child(which):
    """G15: the reference's own Trainer.train() on the tiny dataset, everything it draws and computes recorded."""
    import shutil
    import torch
    repo = os.path.join(HERE, "..")
    sys.path.insert(0, os.path.join(repo, "tests"))
    import helpers as H
    tmp = "/tmp/mmssl_golden_bltr_%s/" % which
    shutil.rmtree(tmp, ignore_errors=True)
    H.write_dataset_dir(tmp)                      # the fixture's dataset, byte for byte what the GPU test rebuilds
    root = "/root/reference/%s/codes" % which
    sys.path.insert(0, root)
    argv = ["--data_path", tmp] + TRAINER_ARGV + (["--sparse", "1"] if which == "MICRO" else [])
    sys.argv = ["main.py"] + argv
    _shims(torch, which == "MICRO")
    import main as M
    dg = M.data_generator
    M.set_seed(M.args.seed)
    config = {"n_users": dg.n_users, "n_items": dg.n_items}
    _, norm_adj, _ = dg.get_adj_mat()
    config["norm_adj"] = norm_adj
    tr = M.Trainer(data_config=config)
    npy = lambda t: t.detach().cpu().numpy().copy()           # noqa: E731
    rec = {"argv": np.array(" ".join(argv[2:])), "n_users": dg.n_users, "n_items": dg.n_items}
    na = norm_adj.tocoo()
    rec["norm_adj_row"], rec["norm_adj_col"], rec["norm_adj_val"] = na.row.astype(np.int64), na.col.astype(np.int64), na.data.astype(np.float32)
    for k, v in tr.model.state_dict().items():
        rec["m0." + k] = npy(v)
    batches, losses, cls, evals = [], [], [], []
    sample0, bpr0, test0 = dg.sample, tr.bpr_loss, tr.test

    def sample():
        out = sample0()
        batches.append([np.asarray(x, dtype=np.int64) for x in out])
        return out

    def bpr_loss(u, p, n):
        out = bpr0(u, p, n)
        losses.append((float(out[0]), float(out[1])))
        return out

    def test(users, is_val):
        ret = test0(users, is_val)
        evals.append((bool(is_val), np.asarray(users, dtype=np.int64), {k: np.asarray(v, dtype=np.float64) for k, v in ret.items()}))
        return ret
    dg.sample, tr.bpr_loss, tr.test = sample, bpr_loss, test
    if which == "MICRO":
        cl0 = tr.model.batched_contrastive_loss

        def cl(z1, z2, *a, **k):
            out = cl0(z1, z2, *a, **k)
            cls.append(float(out))
            return out
        tr.model.batched_contrastive_loss = cl
    tr.train()
    dg.sample, tr.bpr_loss, tr.test = sample0, bpr0, test0
    n = len(batches)
    assert n == len(losses) == 21 and (which != "MICRO" or len(cls) == 2 * n)
    rec["n_batches"] = n
    for b in range(n):
        for nm, x in zip(("users", "pos", "neg"), batches[b]):
            rec["b%d.%s" % (b, nm)] = x
        rec["b%d.mf" % b], rec["b%d.emb" % b] = np.float64(losses[b][0]), np.float64(losses[b][1])
        if which == "MICRO":
            rec["b%d.cl" % b] = np.float64(np.float32(np.float32(cls[2 * b]) + np.float32(cls[2 * b + 1])) * np.float32(M.args.loss_ratio))
    for k, v in tr.model.state_dict().items():
        rec["m1." + k] = npy(v)
    rec["n_evals"] = len(evals)
    for e, (is_val, users, ret) in enumerate(evals):
        rec["e%d.is_val" % e] = np.int64(is_val)
        rec["e%d.users" % e] = users
        for k in ("precision", "recall", "ndcg", "hit_ratio"):
            rec["e%d.%s" % (e, k)] = ret[k]
    tr.model.eval()
    with torch.no_grad():
        outs = tr.model(tr.norm_adj, build_item_graph=True)
    rec["eval.ua"], rec["eval.ia"] = npy(outs[0]), npy(outs[1])
    rec["final_lr"] = np.float64(tr.optimizer.param_groups[0]["lr"])
    np.savez_compressed(os.path.join(OUT, "g15_%s_trainer.npz" % which.lower()), **rec)
    print("G15", which, "batches", n, "evals", len(evals), "first / last mf", losses[0][0], losses[-1][0],
          "recall@20", [float(r[2]["recall"][1]) for r in evals])

    # ---- the PRODUCT loop over the reference's classes (CPU): identical trajectory ---------------------------------
    sys.path.insert(0, repo)
    from mmssl_amd import baselines_main as BM
    import Models

    class CpuTrainer(BM.Trainer):
        def _make_adj(self, a):
            return tr.sparse_mx_to_torch_sparse_tensor(a).float()

        def _make_model(self, img, txt):
            return getattr(Models, which)(self.n_users, self.n_items, self.emb_dim, self.weight_size, self.mess_dropout, img, txt)

        def _make_optimizer(self):
            return torch.optim.Adam(self.model.parameters(), lr=self.lr)

        def _evaluate(self, ua, ia, users, is_val):
            return M.test_torch(ua, ia, users, is_val)

        def _batch_losses(self, outs, users, pos, neg):
            mf, emb, reg = tr.bpr_loss(outs[0][users], outs[1][pos], outs[1][neg])
            cl_ = None
            if which == "MICRO":
                cl_ = (self.model.batched_contrastive_loss(outs[2], outs[4])
                       + self.model.batched_contrastive_loss(outs[3], outs[4])) * self.args.loss_ratio
            got.append((float(mf), float(emb), None if cl_ is None else float(cl_)))
            return mf, emb, reg, cl_
    got = []
    a = BM.parse_args(which.lower(), argv)
    M.set_seed(a.seed)
    ptr = CpuTrainer(data_config=config, args=a, data=dg, device="cpu")
    ptr.model.load_state_dict({k[3:]: torch.from_numpy(v) for k, v in rec.items() if k.startswith("m0.")})
    M.set_seed(a.seed)
    M.Trainer(data_config=config)                  # consume the same random stream the reference's constructor did
    ptr.train()
    assert len(got) == n
    for b in range(n):
        assert got[b][0] == losses[b][0] and got[b][1] == losses[b][1], (b, got[b], losses[b])
        if which == "MICRO":
            assert abs(got[b][2] - float(rec["b%d.cl" % b])) <= 1e-6 * abs(got[b][2]), b
    for k, v in ptr.model.state_dict().items():
        assert np.array_equal(npy(v), rec["m1." + k]), k
    assert abs(ptr.optimizer.param_groups[0]["lr"] - float(rec["final_lr"])) < 1e-15
    vals = [h for h in ptr.history]
    assert len(vals) == sum(1 for e in evals if e[0]), (len(vals), len(evals))
    print("G15", which, ": the product loop over the reference's classes reproduces the trajectory bit for bit")


CASES = {"LATTICE": ("LATTICE", "lightgcn", 0, None, None),
         "MICRO": ("MICRO", "lightgcn", 0, None, None),
         "LATTICE_ngcf": ("LATTICE", "ngcf", 0, None, "g13_lattice_ngcf"),
         "NGCF": ("MICRO", "ngcf", 0, "NGCF", "g13_ngcf"),
         "MICRO_sparse_check": ("MICRO", "lightgcn", 1, None, None),
         "MICRO_ngcf_sparse": ("MICRO", "ngcf", 1, None, "g14_micro_ngcf_sparse")}


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "trainer":
        trainer_child(sys.argv[2])
    elif len(sys.argv) > 1:
        child(*CASES[sys.argv[1]])
    else:
        jobs = [[w] for w in CASES] + [["trainer", "LATTICE"], ["trainer", "MICRO"]]
        for w in jobs:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)] + w, capture_output=True, text=True)
            print(r.stdout[-600:], r.stderr[-1500:] if r.returncode else "")
