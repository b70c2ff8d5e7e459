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
