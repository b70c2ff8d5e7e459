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
    def batch_losses_rows(u, p, n, z_img, z_txt, decay, batch_size, tau):
        mf, emb, _ = O.bpr(u, p, n, decay, batch_size)
        zero = torch.zeros((), dtype=u.dtype)
        return torch.stack([mf, emb, zero, O.infonce(z_img, u, tau), O.infonce(z_txt, u, tau)])

    @staticmethod
    def loss_assemble(terms, w, extra=None, c=0.0):
        total = (terms * w).sum()
        return total if extra is None else total + c * extra

    # ---- non-autograd ("raw") ops used by the fused sharded node ------------------------------------
    @staticmethod
    def spmm_raw(plan, transpose, X, epilogue, Z=None, alpha=0.0, S=None):
        with torch.no_grad():
            Y = O.spmm(plan.AT if transpose else plan.A, X)
            if epilogue == 1:
                Y = torch.softmax(Y, -1)
            elif epilogue in (2, 3):
                Y = Y + alpha * Z
                if epilogue == 3:
                    Y = S * (Y - (Y * S).sum(1, keepdim=True))
            return Y

    @staticmethod
    def linear_raw(F_, W, b, keep, scale):
        with torch.no_grad():
            return OracleBackend.linear(F_, W, b, keep, scale)

    @staticmethod
    def linear_wgrad_raw(gY, keep, scale, F_, W):
        with torch.no_grad():
            if keep is not None:
                gY = gY * keep.to(gY.dtype) * scale
            return gY, gY.t() @ F_, gY.sum(0)

    @staticmethod
    def combine_fwd(layers, inv, A, B, r):
        with torch.no_grad():
            out = inv * torch.stack(list(layers)).sum(0) + r * F.normalize(A) + r * F.normalize(B)
            return out, (A ** 2).sum() + (B ** 2).sum()

    @staticmethod
    def combine_bwd(A, B, G, r, inv, c_dev, c_scale, want_gL):
        outs = []
        for X in (A, B):
            with torch.enable_grad():          # we are inside a custom Function's backward
                x = X.detach().clone().requires_grad_(True)
                y = r * F.normalize(x)
                (gx,) = torch.autograd.grad(y, x, G)
            if c_dev is not None:
                gx = gx + (c_scale * float(c_dev)) * X
            outs.append(gx)
        return outs[0], outs[1], (inv * G if want_gL else None)

    @staticmethod
    def softmax_rows_bwd(Y, gY, scale=1.0):
        with torch.no_grad():
            return scale * Y * (gY - (gY * Y).sum(1, keepdim=True))
