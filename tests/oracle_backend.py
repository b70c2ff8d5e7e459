"""CPU compute backend for mmssl_amd.dist built from the oracle (TEST-ONLY): lets the row-sharding /
collective logic run under gloo without a GPU. The product backend is dist.HipBackend."""
import scipy.sparse as sp
import torch
import torch.nn.functional as F

import mmssl_oracle as O


class _Graph:
    def __init__(self, csr):
        csr = sp.csr_matrix(csr)
        self.shape = csr.shape
        self.nnz = csr.nnz
        self.A = O.to_torch_sparse(csr)
        self.AT = O.to_torch_sparse(csr.T.tocsr())


class OracleBackend:
    EPI_NONE, EPI_SOFTMAX = 0, 1
    make_graph = _Graph

    @staticmethod
    def dropout_masks(count, rows, cols, p, device):
        return torch.empty((count, rows, cols), dtype=torch.uint8, device=device).bernoulli_(1.0 - p)

    @staticmethod
    def spmm(plan, X, epilogue=0, transpose=False):
        Y = O.spmm(plan.AT if transpose else plan.A, X)
        return torch.softmax(Y, -1) if epilogue else Y

    @staticmethod
    def l2norm_rows(X, base=None, alpha=1.0):
        y = alpha * F.normalize(X, p=2, dim=1)
        return y if base is None else base + y

    @staticmethod
    def linear(F_, W, b=None, keep=None, scale=1.0):
        y = F.linear(F_, W, b)
        return y if keep is None else y * keep.to(y.dtype) * scale

    @staticmethod
    def bpr(u, p, n, decay, batch_size):
        mf, emb, _ = O.bpr(u, p, n, decay, batch_size)
        return mf, emb

    @staticmethod
    def infonce(z1, z2, tau):
        return O.infonce(z1, z2, tau)

    @staticmethod
    def sumsq(x):
        return (x ** 2).sum()

    @staticmethod
    def batch_losses_rows(u, ia, z_img, z_txt, decay, batch_size, tau):
        B = u.shape[0]
        p, n = ia[:B], ia[B:]
        mf, emb, _ = O.bpr(u, p, n, decay, batch_size)
        zero = torch.zeros((), dtype=u.dtype)
        return torch.stack([mf, emb, zero, O.infonce(z_img, u, tau), O.infonce(z_txt, u, tau)])

    @staticmethod
    def loss_assemble(terms, w, extra=None, c=0.0):
        total = (terms * w).sum()
        return total if extra is None else total + c * extra

    # ---- non-autograd ("raw") ops used by the fused sharded node ------------------------------------
    @staticmethod
    def spmm_raw(plan, transpose, X, epilogue, Z=None, alpha=0.0, S=None, out=None):
        with torch.no_grad():
            Y = O.spmm(plan.AT if transpose else plan.A, X)
            if epilogue == 1:
                Y = torch.softmax(Y, -1)
            elif epilogue in (2, 3):
                Y = Y + alpha * Z
                if epilogue == 3:
                    Y = S * (Y - (Y * S).sum(1, keepdim=True))
            if out is not None:              # a column chunk of a wider table (the item-side node's lanes)
                out.copy_(Y)
                return out
            return Y

    @staticmethod
    def softmax_rows(X):
        return torch.softmax(X, -1)

    @staticmethod
    def softmax_rows_(X):
        with torch.no_grad():
            return X.copy_(torch.softmax(X, -1))

    # ---- the packed node (dist._ShardedHotForward): modalities side by side, 64-wide each ---------------------------
    EPI_AXPY, EPI_AXPY_SOFTMAX_BWD = 2, 3

    @staticmethod
    def packed_supported(feat_dims, rows, d):
        return True

    @staticmethod
    def proj_forward(Fs, Ws, bs, keep, scale, draw_p=0.0, external_tick=False):
        with torch.no_grad():
            if keep is None and draw_p > 0.0:
                keep = OracleBackend.dropout_masks(len(Fs), Fs[0].shape[0], Ws[0].shape[0], draw_p, Fs[0].device)
            cols = []
            for k, (F_, W, b) in enumerate(zip(Fs, Ws, bs)):
                cols.append(OracleBackend.linear(F_, W, b, None if keep is None else keep[k], scale))
            return torch.cat(cols, 1), keep

    @staticmethod
    def proj_wgrad(G, Fs, want_bias):
        with torch.no_grad():
            dm = G.shape[1] // len(Fs)
            gW = [G[:, k * dm:(k + 1) * dm].t() @ F_ for k, F_ in enumerate(Fs)]
            gb = [G[:, k * dm:(k + 1) * dm].sum(0) for k in range(len(Fs))] if want_bias else None
            return gW, gb

    @staticmethod
    def _fuse_side(layers, Mod, inv, nm, r):
        dm = Mod.shape[1] // nm
        out = inv * torch.stack(list(layers)).sum(0)
        for m in range(nm):
            out = out + r * F.normalize(Mod[:, m * dm:(m + 1) * dm])
        return out

    @staticmethod
    def fuse_fwd(us, MU, its, MI, inv, nm, r):
        with torch.no_grad():
            return (OracleBackend._fuse_side(us, MU, inv, nm, r), OracleBackend._fuse_side(its, MI, inv, nm, r),
                    (MU ** 2).sum() + (MI ** 2).sum())

    @staticmethod
    def fuse_bwd(MU, Gu, G_MU, MI, Gi, G_MI, nm, r, inv, g_ss):
        outs = []
        for Mod, G, Gx in ((MU, Gu, G_MU), (MI, Gi, G_MI)):
            dm = Mod.shape[1] // nm
            cols = []
            for m in range(nm):
                with torch.enable_grad():          # we are inside a custom Function's backward
                    x = Mod[:, m * dm:(m + 1) * dm].detach().clone().requires_grad_(True)
                    (gx,) = torch.autograd.grad(r * F.normalize(x), x, G)
                cols.append(gx)
            gMod = torch.cat(cols, 1)
            if g_ss is not None:
                gMod = gMod + (2.0 * float(g_ss)) * Mod
            if Gx is not None:
                gMod = gMod + Gx
            outs.append(gMod)
        return outs[0], inv * Gu, outs[1]

    @staticmethod
    def mask_packed(G, keep, dm, scale):
        with torch.no_grad():
            nm = G.shape[1] // dm
            k = torch.cat([keep[m] for m in range(nm)], 1).to(G.dtype)
            return G * k * scale

    @staticmethod
    def softmax_rows_bwd(Y, gY, scale=1.0):
        with torch.no_grad():
            return scale * Y * (gY - (gY * Y).sum(1, keepdim=True))
