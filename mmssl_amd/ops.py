"""torch.autograd.Function wrappers over the C ABI (include/mmssl_hip.h).

Every op launches hand-written HIP kernels from libmmssl_hip.so on torch's current stream;
torch is only the allocator / stream / autograd tape. No op has an eager fallback: a CPU
tensor or a missing library raises.

Reference call sites are listed per op (paths relative to /root/reference/MMSSL/).
"""
import ctypes as _ct

import torch

from . import _lib
from .graph import GraphPlan

EPI_NONE = 0
EPI_SOFTMAX = 1
EPI_AXPY = 2
EPI_AXPY_SOFTMAX_BWD = 3
# launch accounting for bench.py (edge.layers = nonzeros of every SpMM launch)
# One edge.layer = one nonzero of one SpMM launch doing a d-wide fp32 multiply-add with d = the embedding width
# (SURVEY 8d); a launch over w-wide rows (the packed modal chain: w = 2 d) does w / d of them per nonzero. `unit_d` = d
# (None: every launch counts its nonzeros once).
STATS = {"enabled": False, "spmm_launches": 0, "edge_layers": 0, "spmm_bytes": 0, "unit_d": None}


def _count_spmm(plan, rows, d):
    STATS["spmm_launches"] += 1
    # a launch over a column CHUNK (width d < unit_d: the sharded step's chunked propagation) counts its fraction
    # (a float sum of exact binary fractions; readers round it)
    STATS["edge_layers"] += plan.nnz * ((d / STATS["unit_d"]) if STATS["unit_d"] else 1)
    STATS["spmm_bytes"] += plan.nnz * (8 + 4 * d) + rows * 4 * d + (rows + 1) * 4
_NORM_EPS = 1e-12        # F.normalize default eps


def _chk(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise _lib.MmsslError("%s must be a CUDA (HIP) tensor: this package has no CPU path" % name)
    if t.dtype != torch.float32:
        raise _lib.MmsslError("%s must be float32 (the reference computes in fp32)" % name)
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------------
# SpMM                                    Models.py:69-73 (mm), :177-186, :201-211
# ---------------------------------------------------------------------------------------
def _spmm_raw(plan, transpose, X, epilogue, Z=None, alpha=0.0, S=None, out=None):
    """Y = op(A) . X with a store epilogue. X, Z and `out` may be COLUMN CHUNKS of wider row-major tables (views with
    stride(1) == 1 and a row pitch > their width: mmssl_spmm_ld_f32, plain product or AXPY; Z must have out's pitch)."""
    rows = plan.shape[1] if transpose else plan.shape[0]
    cols = plan.shape[0] if transpose else plan.shape[1]
    if X.dim() != 2 or X.shape[0] != cols:
        raise _lib.MmsslError("spmm: X has shape %s, expected [%d, d]" % (tuple(X.shape), cols))
    d = X.shape[1]
    if STATS["enabled"]:
        _count_spmm(plan, rows, d)
    if out is not None and (tuple(out.shape) != (rows, d) or out.dtype != torch.float32 or out.device != X.device):
        raise _lib.MmsslError("spmm: `out` must be a [%d, %d] fp32 tensor on X's device" % (rows, d))
    pitched = [t for t in (X, out, Z) if t is not None and t.shape[0] > 1 and t.stride(0) != t.shape[1]]
    if pitched:
        if epilogue not in (EPI_NONE, EPI_AXPY) or S is not None or any(t.stride(1) != 1 for t in pitched):
            raise _lib.MmsslError("spmm: column-chunk operands take the plain product or the AXPY epilogue only")
        Y = out if out is not None else torch.empty((rows, d), dtype=torch.float32, device=X.device)
        ldx = X.stride(0) if X.shape[0] > 1 else d
        ldy = Y.stride(0) if rows > 1 else d
        if Z is not None and rows > 1 and Z.stride(0) != ldy:
            raise _lib.MmsslError("spmm: Z must have the row pitch of the output")
        ws = plan.workspace(transpose, d)
        rc = _lib.lib().mmssl_spmm_ld_f32(plan.handle, int(transpose), _ptr(X), ldx, d, _ptr(Y), ldy, epilogue, _ptr(Z),
                                          float(alpha), _ptr(ws), ws.numel() * 4, _lib.stream_ptr())
        _lib.check(rc, "mmssl_spmm_ld_f32")
        return Y
    Y = out if out is not None else torch.empty((rows, d), dtype=torch.float32, device=X.device)
    ws = plan.workspace(transpose, d)
    rc = _lib.lib().mmssl_spmm_ex_f32(plan.handle, int(transpose), _ptr(X), d, _ptr(Y), epilogue, _ptr(Z),
                                      float(alpha), _ptr(S), _ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "mmssl_spmm_ex_f32")
    return Y


class _Spmm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, plan, transpose, epilogue):
        X = _chk(X, "X")
        Y = _spmm_raw(plan, transpose, X, epilogue)
        ctx.plan, ctx.transpose, ctx.epilogue = plan, transpose, epilogue
        if epilogue == EPI_SOFTMAX:
            ctx.save_for_backward(Y)
        return Y

    @staticmethod
    def backward(ctx, gY):
        gY = _chk(gY, "gY")
        if ctx.epilogue == EPI_SOFTMAX:
            (Y,) = ctx.saved_tensors
            gY = softmax_rows_bwd(Y, gY)
        gX = _spmm_raw(ctx.plan, not ctx.transpose, gY, EPI_NONE)   # gradX = A^T . gradY
        return gX, None, None, None


def spmm(plan, X, epilogue=EPI_NONE, transpose=False):
    """Y = A @ X (or A^T @ X), optionally with the row softmax fused into the store."""
    if not hasattr(plan, "handle"):
        raise _lib.MmsslError("spmm expects a GraphPlan (see mmssl_amd.graph.as_plan)")
    return _Spmm.apply(X, plan, bool(transpose), int(epilogue))


# ---------------------------------------------------------------------------------------
# row kernels            F.normalize Models.py:196-197,217-218; main.py:212-213
# ---------------------------------------------------------------------------------------
def softmax_rows(X, out=None):
    """softmax over the features of every row (the MMSSL_EPI_SOFTMAX store epilogue as a launch of its own; in place with
    out=X)."""
    Y = torch.empty_like(X) if out is None else out
    rc = _lib.lib().mmssl_softmax_rows_f32(_ptr(X), X.shape[0], X.shape[1], _ptr(Y), _lib.stream_ptr())
    _lib.check(rc, "mmssl_softmax_rows_f32")
    return Y


class _SoftmaxRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X):
        Y = softmax_rows(_chk(X, "X"))
        ctx.save_for_backward(Y)
        return Y

    @staticmethod
    def backward(ctx, gY):
        (Y,) = ctx.saved_tensors
        return softmax_rows_bwd(Y, _chk(gY, "gY"))


def softmax_rows_fn(X):
    """Differentiable row softmax (torch.softmax(X, -1), Models.py:203-204) on the row kernels."""
    return _SoftmaxRows.apply(X)


def softmax_rows_bwd(Y, gY, scale=1.0):
    gX = torch.empty_like(Y)
    rc = _lib.lib().mmssl_softmax_rows_bwd_f32(_ptr(Y), _ptr(gY), float(scale), Y.shape[0], Y.shape[1], _ptr(gX),
                                               _lib.stream_ptr())
    _lib.check(rc, "mmssl_softmax_rows_bwd_f32")
    return gX


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, base, alpha):
        X = _chk(X, "X")
        if base is not None:
            base = _chk(base, "base")
            if base.shape != X.shape:
                raise _lib.MmsslError("l2norm: base shape mismatch")
        Y = torch.empty_like(X)
        rc = _lib.lib().mmssl_l2norm_rows_f32(_ptr(X), _ptr(base), float(alpha), X.shape[0], X.shape[1],
                                              _NORM_EPS, _ptr(Y), _lib.stream_ptr())
        _lib.check(rc, "mmssl_l2norm_rows_f32")
        ctx.save_for_backward(X)
        ctx.alpha = float(alpha)
        ctx.has_base = base is not None
        return Y

    @staticmethod
    def backward(ctx, gY):
        (X,) = ctx.saved_tensors
        gY = _chk(gY, "gY")
        gX = None
        if ctx.needs_input_grad[0]:
            gX = torch.empty_like(X)
            rc = _lib.lib().mmssl_l2norm_rows_bwd_f32(_ptr(X), _ptr(gY), ctx.alpha, X.shape[0], X.shape[1],
                                                      _NORM_EPS, _ptr(gX), _lib.stream_ptr())
            _lib.check(rc, "mmssl_l2norm_rows_bwd_f32")
        return gX, (gY if ctx.has_base and ctx.needs_input_grad[1] else None), None


def l2norm_rows(X, base=None, alpha=1.0):
    """alpha * X / max(||X||_2, 1e-12) row-wise (+ base): F.normalize(X, p=2, dim=1) with the
    reference's surrounding `base + rate * normalize(x)` fused in."""
    return _L2Norm.apply(X, base, alpha)


class _SumSq(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X):
        X = _chk(X, "X")
        out = torch.empty((), dtype=torch.float32, device=X.device)
        nb = _lib.lib().mmssl_sumsq_workspace_bytes(X.numel())
        ws = torch.empty(nb // 4, dtype=torch.float32, device=X.device)
        rc = _lib.lib().mmssl_sumsq_f32(_ptr(X), X.numel(), _ptr(out), _ptr(ws), nb, _lib.stream_ptr())
        _lib.check(rc, "mmssl_sumsq_f32")
        ctx.save_for_backward(X)
        return out

    @staticmethod
    def backward(ctx, g):
        (X,) = ctx.saved_tensors
        return X * (2.0 * g)


def sumsq(X):
    """(X**2).sum() with a deterministic two-stage reduction (main.py:252-257, 503)."""
    return _SumSq.apply(X)


# ---------------------------------------------------------------------------------------
# modality projection          nn.Linear + nn.Dropout, Models.py:28-29,54,173-174
# ---------------------------------------------------------------------------------------
def _linear_raw(F_, W, b, keep, scale):
    M, K = F_.shape
    N = W.shape[0]
    Y = torch.empty((M, N), dtype=torch.float32, device=F_.device)
    nb = _lib.lib().mmssl_linear_workspace_bytes(M, K, N)
    ws = torch.empty(max(nb // 4, 4), dtype=torch.float32, device=F_.device)
    rc = _lib.lib().mmssl_linear_f32(_ptr(F_), _ptr(W), _ptr(b), _ptr(keep), float(scale), M, K, N, _ptr(Y),
                                     _ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "mmssl_linear_f32")
    return Y


# Dropout keep-masks: one Philox launch for any number of equally shaped masks (nn.Dropout, Models.py:54).
_RNG_STATE = {}


def _rng_state(device):
    key = (device.type, device.index)
    st = _RNG_STATE.get(key)
    if st is None:
        st = torch.tensor([torch.initial_seed() & 0x7FFFFFFFFFFFFFFF, 0, 0], dtype=torch.int64, device=device)
        _RNG_STATE[key] = st
    return st


def seed_dropout(seed, device=None):
    """Reset the mask generator (all devices seen so far, or `device`): masks are a pure function of
    (seed, number of launches since the reset). Unseeded, the state starts from torch.initial_seed()."""
    val = torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0, 0], dtype=torch.int64)
    if device is not None:
        device = torch.device(device)
        _RNG_STATE[(device.type, device.index)] = val.to(device)
        return
    for key in list(_RNG_STATE):
        _RNG_STATE[key].copy_(val)


def tick_rng(device):
    """Advance the mask generator's launch counter by one (what a dropout_masks launch does by itself; callers that
    draw masks inside another kernel - hotnode - or own the step's counters advance it explicitly)."""
    st = _rng_state(torch.device(device))
    rc = _lib.lib().mmssl_tick_u64(st.data_ptr() + 8, _lib.stream_ptr())
    _lib.check(rc, "mmssl_tick_u64")


def dropout_masks(count, rows, cols, p, device, external_tick=False):
    """uint8 [count, rows, cols], 1 = keep with probability 1-p; the generator state lives on the device and
    advances by itself (external_tick: the caller advances it), so a captured step draws fresh masks on every replay."""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.MmsslError("dropout_masks: needs a CUDA (HIP) device")
    n = count * rows * cols
    pad = (-n) % 4
    buf = torch.empty(n + pad, dtype=torch.uint8, device=device)
    rc = _lib.lib().mmssl_dropout_mask_ex_u8(_ptr(_rng_state(device)), float(p), n + pad, _ptr(buf),
                                             1 if external_tick else 0, _lib.stream_ptr())
    _lib.check(rc, "mmssl_dropout_mask_ex_u8")
    return buf[:n].view(count, rows, cols)


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, F_, W, b, keep, scale):
        F_, W = _chk(F_, "F"), _chk(W, "W")
        b = _chk(b, "b") if b is not None else None
        M, K = F_.shape
        N = W.shape[0]
        if W.shape[1] != K:
            raise _lib.MmsslError("linear: W is %s, expected [N, %d]" % (tuple(W.shape), K))
        if keep is not None:
            if keep.dtype != torch.uint8 or tuple(keep.shape) != (M, N) or not keep.is_cuda:
                raise _lib.MmsslError("linear: keep mask must be uint8 [M, N] on the GPU")
            keep = keep.contiguous()
        Y = _linear_raw(F_, W, b, keep, scale)
        ctx.save_for_backward(F_, W, keep)
        ctx.scale = float(scale)
        ctx.has_bias = b is not None
        return Y

    @staticmethod
    def backward(ctx, gY):
        F_, W, keep = ctx.saved_tensors
        gY = _chk(gY, "gY")
        gY_in = gY
        gY, gW, gb = _linear_wgrad_raw(gY, keep, ctx.scale, F_, W)
        gF = None
        if ctx.needs_input_grad[0]:
            if gY is None:                 # the register-direct kernel masks on its fragments: no masked copy exists
                gY = gY_in
                if keep is not None:
                    gY = torch.empty_like(gY_in)
                    rc = _lib.lib().mmssl_mask_scale_f32(_ptr(gY_in), _ptr(keep), ctx.scale, gY_in.numel(), _ptr(gY),
                                                         _lib.stream_ptr())
                    _lib.check(rc, "mmssl_mask_scale_f32")
            # gF = gY @ W  ([M,N] x [N,K]); used by the small modality-fusion product (K = d).
            # The raw feature matrices are constants in the reference (Models.py:46-47).
            if F_.shape[1] > 256 and W.shape[0] % 32 != 0:
                raise _lib.MmsslError("linear: the input gradient of a wide input (K > 256) needs N % 32 == 0")
            gF = _linear_raw(gY, W.t().contiguous(), None, None, 1.0)      # gY already masked
        return gF, gW, (gb if ctx.has_bias else None), None, None


def _linear_wgrad_raw(gY, keep, scale, F_, W):
    """(masked gY or None, gW, gb) for Y = dropout(F W^T + b): dropout backward, then the wgrad GEMM."""
    M, K = F_.shape
    N = W.shape[0]
    fused = keep is not None and _lib.lib().mmssl_linear_wgrad_fuses_mask(M, K, N) == 1
    if keep is not None and not fused:   # register-staged kernel: one dropout-backward pass first
        gYm = torch.empty_like(gY)
        rc = _lib.lib().mmssl_mask_scale_f32(_ptr(gY), _ptr(keep), float(scale), gY.numel(), _ptr(gYm), _lib.stream_ptr())
        _lib.check(rc, "mmssl_mask_scale_f32")
        gY = gYm
    gW = torch.empty_like(W)
    gb = torch.empty(N, dtype=torch.float32, device=W.device)
    nb = _lib.lib().mmssl_linear_wgrad_workspace_bytes(M, K, N)
    ws = torch.empty(max(nb // 4, 4), dtype=torch.float32, device=W.device)
    # register-direct kernel (fused): dropout backward + bias gradient on the loaded fragments
    rc = _lib.lib().mmssl_linear_wgrad_f32(_ptr(gY), _ptr(keep) if fused else None, float(scale) if fused else 1.0, _ptr(F_),
                                           M, K, N, _ptr(gW), _ptr(gb), _ptr(ws), nb, _lib.stream_ptr())
    _lib.check(rc, "mmssl_linear_wgrad_f32")
    return (None if fused else gY), gW, gb


# ---------------------------------------------------------------------------------------
# grouped projection: every modality of one step in ONE stream-K launch (csrc/projection.hip)
# ---------------------------------------------------------------------------------------
def _c_int_arr(vals):
    return (_ct.c_int * len(vals))(*[int(v) for v in vals])


def _c_ptr_arr(tensors):
    return (_ct.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


# The grouped projection runs in SPLIT PRECISION by default (csrc/projection.hip "projx": exact 3-way bf16 cut of every fp32
# value, six partial products on the bf16 matrix pipe, fp32 accumulation - fp32-accurate, bound by the feature stream instead
# of the fp32 MFMA rate). False = the fp32-MFMA kernels (A/B runs: bench.py --proj f32).
PROJ_SPLIT = True
def projx_supported(Ks, M, N):
    return len(Ks) >= 1 and _lib.lib().mmssl_projx_supported(len(Ks), _c_int_arr(Ks), int(M), int(N)) == 1


def proj_images(F, transposed=False):
    """The tile-major image of the constant feature matrix F [M, K] (reference Models.py:46-47) that the split-precision
    projection streams: of F itself (forward) or of F^T (weight gradient). Built once per matrix by mmssl_projx_pack_f32 and
    kept ON the tensor object (`F._mmssl_proj_images`): the images live exactly as long as F does - whoever holds F (the
    model, and through it a step object and its captured hipGraph, whose nodes carry the images' addresses) holds them, and
    nothing else does (until round 6 a global 8-entry FIFO owned them: an eviction could free an image a live graph still
    read, and entries outlived their models). An in-place change of F (its version counter) rebuilds the images."""
    e = getattr(F, "_mmssl_proj_images", None)
    if e is None or e[0] != F._version or e[3] != (F.data_ptr(), tuple(F.shape), tuple(F.stride())):
        e = [F._version, None, None, (F.data_ptr(), tuple(F.shape), tuple(F.stride()))]
        F._mmssl_proj_images = e
    k = 2 if transposed else 1
    if e[k] is None:
        M, K = F.shape
        rows, red = (K, M) if transposed else (M, K)
        n = _lib.lib().mmssl_projx_image_floats(rows, red)
        img = torch.empty(n, dtype=torch.float32, device=F.device)
        rc = _lib.lib().mmssl_projx_pack_f32(_ptr(F), M, K, F.stride(0), 1 if transposed else 0, _ptr(img), _lib.stream_ptr())
        _lib.check(rc, "mmssl_projx_pack_f32")
        e[k] = img
    return e[k]


_CU_COUNT = {}


def proj_step_blocks(dev):
    """Block count for a projection launch that runs BESIDE other kernels (the step's GCN chain on a side stream): 13/16 of
    the CUs. The split-precision launch is bound by the feature stream, which that many CUs still saturate; under a
    full-width launch a side-stream SpMM took 63 us instead of 11 (profiles/r05/projx_blocks.txt)."""
    key = torch.device(dev).index
    n = _CU_COUNT.get(key)
    if n is None:
        n = _CU_COUNT[key] = torch.cuda.get_device_properties(dev).multi_processor_count
    return max(1, (13 * n) // 16)


def _projx_ws(n, Ks, M, N, wgrad, dev, blocks=0):
    nb = _lib.lib().mmssl_projx_workspace_bytes(n, _c_int_arr(Ks), M, N, int(wgrad), int(blocks))
    if nb == 0:
        raise _lib.MmsslError("split-precision projection: unsupported modality list K=%s M=%d N=%d" % (Ks, M, N))
    ws = torch.empty(nb // 4 + 64, dtype=torch.float32, device=dev)
    off = (-ws.data_ptr() % 256) // 4             # the entry points want a 256-byte aligned workspace
    return ws[off:], nb


class WeightPlanes:
    """The bf16 planes of the projection weights as a step-owned image (csrc/projection.hip projx_wsplit_kernel), made right
    BEHIND the optimiser's update - at the end of a step - so that the next forward starts with its main kernel instead of a
    14 us split launch on the critical path. The image is only used while it is provably that of the current weights: the
    key holds every weight's (address, torch version counter, `_mmssl_serial`); torch's in-place writers (copy_,
    load_state_dict, torch optimisers) bump the version, optim.FusedAdamW.step bumps `_mmssl_serial` of what it updates, and
    the fused weight-gradient epilogue - the one writer that bumps nothing - is followed by refresh()."""

    def __init__(self):
        self.buf, self.key, self.shape = None, None, None

    @staticmethod
    def _key(Ws):
        return tuple((w.data_ptr(), w._version, getattr(w, "_mmssl_serial", 0)) for w in Ws)

    def refresh(self, Ws):
        """Split the CURRENT weights into the image (one launch on the current stream)."""
        Ks = [int(w.shape[1]) for w in Ws]
        n = len(Ws)
        nb = _lib.lib().mmssl_projx_wimg_bytes(n, _c_int_arr(Ks))
        if nb == 0:
            self.key = None
            return False
        shape = (tuple(Ks), Ws[0].device)
        if self.buf is None or self.shape != shape:
            raw = torch.empty(nb // 4 + 64, dtype=torch.float32, device=Ws[0].device)
            off = (-raw.data_ptr() % 256) // 4
            self.buf, self.shape = raw[off:off + nb // 4], shape          # exactly the image (256-byte aligned)
        rc = _lib.lib().mmssl_projx_wsplit_f32(n, _c_ptr_arr(Ws), _c_int_arr(Ks), _ptr(self.buf), _lib.stream_ptr())
        _lib.check(rc, "mmssl_projx_wsplit_f32")
        self.key = self._key(Ws)
        return True

    def image_for(self, Ws):
        """The image if it is the current weights', else None."""
        return self.buf if (self.key is not None and self.key == self._key(Ws)) else None

    def writable_for(self, Ws):
        """True when the image exists with these weights' shapes AND was last made for these very tensors: the fused
        weight-gradient epilogue (proj_wgrad(..., planes=self)) may then rewrite the planes of what it updates - it touches
        only the entries of real weights, so the zero padding of the last slice must already be in place."""
        if self.buf is None or self.key is None:
            return False
        return self.shape == (tuple(int(w.shape[1]) for w in Ws), Ws[0].device) and \
            tuple(k[0] for k in self.key) == tuple(w.data_ptr() for w in Ws)

    def mark_current(self, Ws):
        """The planes were rewritten by the kernel that updated the weights: the image is the current weights' again."""
        self.key = self._key(Ws)


def proj_supported(Ks, M, N, wgrad=False):
    """True when the grouped projection runs this modality list: N == 64 and, split precision (default), K % 4 == 0; on the
    fp32-MFMA kernels K % 32 == 0 forward, K % 4 == 0 weight gradient."""
    if len(Ks) < 1:
        return False
    if PROJ_SPLIT and projx_supported(Ks, M, N):
        return True
    return _lib.lib().mmssl_proj_supported(len(Ks), _c_int_arr(Ks), int(M), int(N), int(bool(wgrad))) == 1


def proj_forward(Fs, Ws, bs, keep=None, draw=None, scale=1.0, blocks=0, planes=None):
    """Y [M, 64 * n] = the projections dropout(F_g W_g^T + b_g) of all modalities side by side, one launch + one
    epilogue launch. `planes` (a WeightPlanes, split precision only): use its image of the weights when it is current. `blocks` (split precision only): the launch's block count, 0 = one per CU (see proj_step_blocks). `keep`: uint8 [n, M, 64] given masks; `draw` = (p, device rng state tensor): the masks are drawn in
    the epilogue (same bytes as ops.dropout_masks(n, M, 64, p) at the same generator state) and returned; the caller
    advances the generator (dropout_masks's external-tick contract). Returns (Y, keep or None)."""
    n = len(Fs)
    M, N = Fs[0].shape[0], Ws[0].shape[0]
    Ks = [f.shape[1] for f in Fs]
    dev = Fs[0].device
    Y = torch.empty((M, N * n), dtype=torch.float32, device=dev)
    split = PROJ_SPLIT and projx_supported(Ks, M, N)
    if split:
        imgs = [proj_images(f) for f in Fs]
        ws, nb = _projx_ws(n, Ks, M, N, 0, dev, blocks)
    else:
        nb = _lib.lib().mmssl_proj_workspace_bytes(n, _c_int_arr(Ks), M, N, 0)
        if nb == 0:
            raise _lib.MmsslError("proj_forward: unsupported modality list K=%s M=%d N=%d" % (Ks, M, N))
        ws = torch.empty(nb // 4 + 4, dtype=torch.float32, device=dev)
    keep_out, rng, p = None, None, 0.0
    if draw is not None:
        p, rng = float(draw[0]), draw[1]
        keep_out = torch.empty((n, M, N), dtype=torch.uint8, device=dev)
    elif keep is not None:
        if keep.dtype != torch.uint8 or tuple(keep.shape) != (n, M, N) or not keep.is_contiguous():
            raise _lib.MmsslError("proj_forward: keep must be a contiguous uint8 [n, M, 64] tensor")
    if split:
        wimg = planes.image_for(Ws) if planes is not None else None       # WeightPlanes: made behind the last update
        if wimg is not None:
            rc = _lib.lib().mmssl_projx_fwd_img_f32(n, _c_ptr_arr(imgs), _ptr(wimg), _c_ptr_arr(bs), _c_int_arr(Ks), M, N,
                                                    _ptr(keep), _ptr(keep_out), _ptr(rng), p, float(scale), _ptr(Y), N * n,
                                                    int(blocks), _ptr(ws), nb, _lib.stream_ptr())
            _lib.check(rc, "mmssl_projx_fwd_img_f32")
            return Y, (keep_out if draw is not None else keep)
        rc = _lib.lib().mmssl_projx_fwd_f32(n, _c_ptr_arr(imgs), _c_ptr_arr(Ws), _c_ptr_arr(bs), _c_int_arr(Ks), M, N,
                                            _ptr(keep), _ptr(keep_out), _ptr(rng), p, float(scale), _ptr(Y), N * n, int(blocks),
                                            _ptr(ws), nb, _lib.stream_ptr())
        _lib.check(rc, "mmssl_projx_fwd_f32")
        return Y, (keep_out if draw is not None else keep)
    rc = _lib.lib().mmssl_proj_fwd_f32(n, _c_ptr_arr(Fs), _c_ptr_arr(Ws), _c_ptr_arr(bs), _c_int_arr(Ks), M, N, _ptr(keep),
                                       _ptr(keep_out), _ptr(rng), p, float(scale), _ptr(Y), N * n, _ptr(ws),
                                       ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "mmssl_proj_fwd_f32")
    return Y, (keep_out if draw is not None else keep)


def proj_wgrad(G, Fs, want_bias=True, adam=None, blocks=0, planes=None):
    """([gW_g [64, K_g]], [gb_g [64]]) from the ALREADY masked output gradient G [M, 64 * n] (modalities side by side)
    and the feature matrices, one launch + one epilogue launch. `adam` (optim.FusedAdamW.fused_slots): the epilogue also
    applies the AdamW update of the projection weights / biases to the gradient it has just summed; `planes` (a
    WeightPlanes that is writable_for these weights; split precision + adam only): it also rewrites their bf16 planes."""
    n = len(Fs)
    M = Fs[0].shape[0]
    N = G.shape[1] // n
    Ks = [f.shape[1] for f in Fs]
    dev = G.device
    gW = [torch.empty((N, k), dtype=torch.float32, device=dev) for k in Ks]
    gb = [torch.empty(N, dtype=torch.float32, device=dev) for _ in Ks] if want_bias else None
    if PROJ_SPLIT and projx_supported(Ks, M, N):
        imgs = [proj_images(f, transposed=True) for f in Fs]
        ws, nb = _projx_ws(n, Ks, M, N, 1, dev, blocks)
        if adam is None:
            rc = _lib.lib().mmssl_projx_wgrad_f32(n, _ptr(G), G.stride(0), _c_ptr_arr(imgs), _c_int_arr(Ks), M, N,
                                                  _c_ptr_arr(gW), _c_ptr_arr(gb) if gb else None, int(blocks), _ptr(ws), nb,
                                                  _lib.stream_ptr())
            _lib.check(rc, "mmssl_projx_wgrad_f32")
            return gW, gb
        a = adam
        bias = a["b"] if any(t is not None for t in a["b"]) else None
        rc = _lib.lib().mmssl_projx_wgrad_adamw_img_f32(
            n, _ptr(G), G.stride(0), _c_ptr_arr(imgs), _c_int_arr(Ks), M, N, _c_ptr_arr(gW), _c_ptr_arr(gb) if gb else None,
            _c_ptr_arr(a["W"]), _c_ptr_arr(a["mW"]), _c_ptr_arr(a["vW"]), _c_ptr_arr(a["b"]) if bias else None,
            _c_ptr_arr(a["mb"]) if bias else None, _c_ptr_arr(a["vb"]) if bias else None, _ptr(a["state"]), a["lr"],
            a["beta1"], a["beta2"], a["eps"], a["weight_decay"], 1 if a["pre_ticked"] else 0,
            _ptr(planes.buf) if planes is not None else None, int(blocks), _ptr(ws), nb, _lib.stream_ptr())
        _lib.check(rc, "mmssl_projx_wgrad_adamw_img_f32")
        return gW, gb
    nb = _lib.lib().mmssl_proj_workspace_bytes(n, _c_int_arr(Ks), M, N, 1)
    if nb == 0:
        raise _lib.MmsslError("proj_wgrad: unsupported modality list K=%s M=%d N=%d" % (Ks, M, N))
    ws = torch.empty(nb // 4 + 4, dtype=torch.float32, device=dev)
    if adam is None:
        rc = _lib.lib().mmssl_proj_wgrad_f32(n, _ptr(G), G.stride(0), _c_ptr_arr(Fs), _c_int_arr(Ks), M, N, _c_ptr_arr(gW),
                                             _c_ptr_arr(gb) if gb else None, _ptr(ws), ws.numel() * 4, _lib.stream_ptr())
        _lib.check(rc, "mmssl_proj_wgrad_f32")
        return gW, gb
    a = adam
    bias = a["b"] if any(t is not None for t in a["b"]) else None
    rc = _lib.lib().mmssl_proj_wgrad_adamw_f32(
        n, _ptr(G), G.stride(0), _c_ptr_arr(Fs), _c_int_arr(Ks), M, N, _c_ptr_arr(gW), _c_ptr_arr(gb) if gb else None,
        _c_ptr_arr(a["W"]), _c_ptr_arr(a["mW"]), _c_ptr_arr(a["vW"]), _c_ptr_arr(a["b"]) if bias else None,
        _c_ptr_arr(a["mb"]) if bias else None, _c_ptr_arr(a["vb"]) if bias else None, _ptr(a["state"]), a["lr"], a["beta1"],
        a["beta2"], a["eps"], a["weight_decay"], 1 if a["pre_ticked"] else 0, _ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "mmssl_proj_wgrad_adamw_f32")
    return gW, gb


def linear(F_, W, b=None, keep=None, scale=1.0):
    """dropout(F @ W^T + b) on fp32 MFMA; `keep` is a uint8 keep-mask (None = eval mode)."""
    return _Linear.apply(F_, W, b, keep, scale)


# ---------------------------------------------------------------------------------------
# InfoNCE                       Trainer.sim + batched_contrastive_loss, main.py:211-249
# ---------------------------------------------------------------------------------------
def _infonce_fwd_raw(z1, z2, idx, tau, loss=None, log_eps=1e-8):
    n = z1.shape[0] if idx is None else idx.shape[0]
    d = z1.shape[1]
    nb = _lib.lib().mmssl_infonce_workspace_bytes(n, d)
    if nb == 0:
        raise _lib.MmsslError("infonce: unsupported shape n=%d d=%d" % (n, d))
    ws = torch.empty(nb // 4, dtype=torch.float32, device=z1.device)
    if loss is None:
        loss = torch.empty((), dtype=torch.float32, device=z1.device)
    rc = _lib.lib().mmssl_infonce_fwd_eps_f32(_ptr(z1), _ptr(z2), _ptr(idx), n, d, float(tau), float(log_eps), _ptr(loss),
                                              _ptr(ws), nb, _lib.stream_ptr())
    _lib.check(rc, "mmssl_infonce_fwd_eps_f32")
    return loss, ws, n, d


def _infonce_bwd_raw(idx, n, d, tau, g, gz1, gz2, ws):
    rc = _lib.lib().mmssl_infonce_bwd_f32(_ptr(idx), n, d, float(tau), _ptr(g), _ptr(gz1), _ptr(gz2), _ptr(ws),
                                          ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "mmssl_infonce_bwd_f32")


class _InfoNCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z1, z2, idx, tau, log_eps=1e-8):
        z1, z2 = _chk(z1, "z1"), _chk(z2, "z2")
        if z1.dim() != 2 or z2.dim() != 2 or z1.shape[1] != z2.shape[1] or (idx is None and z1.shape != z2.shape):
            raise _lib.MmsslError("infonce: z1/z2 must be [n, d] (or tables + idx)")
        loss, ws, n, d = _infonce_fwd_raw(z1, z2, idx, tau, log_eps=log_eps)
        ctx.save_for_backward(ws, idx)
        ctx.cfg = (n, d, float(tau), z1.shape, z2.shape)
        return loss

    @staticmethod
    def backward(ctx, g):
        ws, idx = ctx.saved_tensors
        n, d, tau, s1, s2 = ctx.cfg
        g = g.contiguous().to(torch.float32)
        alloc = torch.zeros if idx is not None else torch.empty
        gz1 = alloc(s1, dtype=torch.float32, device=ws.device) if ctx.needs_input_grad[0] else None
        gz2 = alloc(s2, dtype=torch.float32, device=ws.device) if ctx.needs_input_grad[1] else None
        if gz1 is not None or gz2 is not None:
            _infonce_bwd_raw(idx, n, d, tau, g, gz1, gz2, ws)
        return gz1, gz2, None, None, None


def infonce(z1, z2, tau=0.5, idx=None, log_eps=1e-8):
    """The reference's batched_contrastive_loss(z1, z2) (its 1024-row blocking is exactly the
    full-matrix formula). With `idx` (int64 [n]) z1/z2 are whole tables and the
    gather table[idx] is fused into the kernels (main.py:411-412 gathers both by `users`).
    `log_eps`: the constant inside the log (1e-8 in the trainer, main.py:244; 0 in Models.py:79-98 and MICRO)."""
    return _InfoNCE.apply(z1, z2, None if idx is None else _idx(idx, "idx", z1.device), tau, float(log_eps))


# ---------------------------------------------------------------------------------------
# BPR                      gathers main.py:368-370 + Trainer.bpr_loss main.py:499-511
# ---------------------------------------------------------------------------------------
def _idx(t, name, device):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t, dtype=torch.int64)
    t = t.to(device=device, dtype=torch.int64)
    return t if t.is_contiguous() else t.contiguous()


class _Bpr(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Eu, Ei, Ein, users, pos, neg, decay, batch_size):
        Eu, Ei = _chk(Eu, "Eu"), _chk(Ei, "Ei")
        Ein = _chk(Ein, "Ei_neg") if Ein is not None else None
        gathered = users is None
        if gathered:
            B = Eu.shape[0]
            if Ein is None or Ei.shape != Eu.shape or Ein.shape != Eu.shape:
                raise _lib.MmsslError("bpr: gathered form needs three [B, d] tensors")
        else:
            B = users.shape[0]
        d = Eu.shape[1]
        out = torch.empty(3, dtype=torch.float32, device=Eu.device)
        nb = _lib.lib().mmssl_bpr_workspace_bytes(B)
        ws = torch.empty(nb // 4, dtype=torch.float32, device=Eu.device)
        rc = _lib.lib().mmssl_bpr_fwd_f32(_ptr(Eu), _ptr(Ei), _ptr(Ein), _ptr(users), _ptr(pos), _ptr(neg), B, d,
                                          float(decay), int(batch_size), _ptr(out), _ptr(ws), nb,
                                          _lib.stream_ptr())
        _lib.check(rc, "mmssl_bpr_fwd_f32")
        ctx.save_for_backward(Eu, Ei, Ein, users, pos, neg)
        ctx.cfg = (B, d, float(decay), int(batch_size), gathered)
        return out

    @staticmethod
    def backward(ctx, g_out):
        Eu, Ei, Ein, users, pos, neg = ctx.saved_tensors
        B, d, decay, batch_size, gathered = ctx.cfg
        g_out = g_out.contiguous().to(torch.float32)
        g_mf, g_emb = g_out[0:1], g_out[1:2]       # views: adjacent floats of one buffer
        gEu = torch.zeros_like(Eu)
        gEi = torch.zeros_like(Ei)
        gEin = torch.zeros_like(Ein) if gathered else None
        rc = _lib.lib().mmssl_bpr_bwd_f32(_ptr(Eu), _ptr(Ei), _ptr(Ein), _ptr(users), _ptr(pos), _ptr(neg), B, d,
                                          decay, batch_size, _ptr(g_mf), _ptr(g_emb), _ptr(gEu), _ptr(gEi),
                                          _ptr(gEin), _lib.stream_ptr())
        _lib.check(rc, "mmssl_bpr_bwd_f32")
        return gEu, gEi, gEin, None, None, None, None, None


def bpr_gather(Eu, Ei, users, pos, neg, decay, batch_size):
    """Fused gather + BPR: (mf_loss, emb_loss) from the full tables and the int64 batch."""
    dev = Eu.device
    out = _Bpr.apply(Eu, Ei, None, _idx(users, "users", dev), _idx(pos, "pos", dev), _idx(neg, "neg", dev),
                     decay, batch_size)
    return out[0], out[1]


def bpr(u, p, n, decay, batch_size):
    """Trainer.bpr_loss on already-gathered [B, d] rows (the reference's signature)."""
    out = _Bpr.apply(u, p, n, None, None, None, decay, batch_size)
    return out[0], out[1]


# ---------------------------------------------------------------------------------------
# u_sim + real-data rows              Trainer.u_sim_calculation main.py:281-298, :349
# ---------------------------------------------------------------------------------------
def _pitch(n, mult=32):
    return (n + mult - 1) // mult * mult


_USIM_WS = {}


def _usim_ws(nbytes, dev):
    """The Gram-matrix workspace of mmssl_usim_rows_f32 (block partials + the float64 matrix), one per stream: calls on
    one stream are ordered, two streams must not share it."""
    key = (str(dev), torch.cuda.current_stream(dev).cuda_stream, int(nbytes))
    ws = _USIM_WS.get(key)
    if ws is None:
        ws = _USIM_WS[key] = torch.empty(int(nbytes) // 4 + 4, dtype=torch.float32, device=dev)
    return ws


def sim_rows(Q, T, qidx=None, mask=None, mask_value=0.0, normalize=False, eps=_NORM_EPS, pitch_mult=1):
    """S[b, j] = <Q[qidx[b]], T[j]> for every row j of T, on the fp32 matrix cores, with the entries of a CSR mask row
    replaced by `mask_value` and (optionally) the rows L2-normalised: the score product of
    Trainer.u_sim_calculation (main.py:283-298) / of the evaluation (utility/batch_test.py:150-152, 91-100) as ONE tile
    kernel + one scale pass (csrc/simtopk.hip) instead of a library GEMM followed by masking / norm passes.
    `mask`: None, a GraphPlan (its CSR pattern, rows indexed by qidx) or a pair of int32 device tensors
    (rowptr, sorted cols). Returns (S [B, n] view of a [B, pitch] buffer, inv_norm [B] or None)."""
    Q, T = _chk(Q, "Q"), _chk(T, "T")
    if Q.dim() != 2 or T.dim() != 2 or Q.shape[1] != T.shape[1]:
        raise _lib.MmsslError("sim_rows: Q [*, d] and T [n, d] expected")
    dev = Q.device
    qidx = _idx(qidx, "qidx", dev)
    B = Q.shape[0] if qidx is None else qidx.shape[0]
    n, d = T.shape
    ld = _pitch(n, pitch_mult)
    L = _lib.lib()
    one_pass = normalize and float(mask_value) == 0.0 and L.mmssl_usim_workspace_bytes(d, n) > 0
    if one_pass:
        # u_sim: the row factors first (Gram matrix of T, float64 quadratic forms), then ONE pass that writes the scaled
        # matrix, pad columns included - no zero fill, no partial sums, no second pass over [B, n]
        out = torch.empty((B, ld), dtype=torch.float32, device=dev)
        inv = torch.empty(B, dtype=torch.float32, device=dev)
        nb = L.mmssl_usim_workspace_bytes(d, n)
        ws = _usim_ws(nb, dev)
        if mask is not None and hasattr(mask, "handle"):
            if qidx is None:
                raise _lib.MmsslError("sim_rows: a plan mask needs qidx (the plan rows of the batch)")
            rc = L.mmssl_graph_usim_rows_f32(mask.handle, _ptr(Q), _ptr(qidx), B, _ptr(T), d, float(eps), _ptr(out), ld,
                                             _ptr(inv), _ptr(ws), nb, _lib.stream_ptr())
            _lib.check(rc, "mmssl_graph_usim_rows_f32")
        else:
            rp, cols = (None, None) if mask is None else mask
            rc = L.mmssl_usim_rows_f32(_ptr(Q), _ptr(qidx), B, _ptr(T), n, d, _ptr(rp), _ptr(cols), float(eps), _ptr(out),
                                       ld, _ptr(inv), _ptr(ws), nb, _lib.stream_ptr())
            _lib.check(rc, "mmssl_usim_rows_f32")
        return (out if ld == n else out[:, :n]), inv
    out = torch.empty((B, ld), dtype=torch.float32, device=dev)       # (the tile kernel writes the pad columns as zeros)
    nparts = _lib.lib().mmssl_sim_rows_parts(n)
    part = torch.empty((B, max(nparts, 1)), dtype=torch.float32, device=dev) if normalize else None
    if mask is not None and hasattr(mask, "handle"):
        if qidx is None:
            raise _lib.MmsslError("sim_rows: a plan mask needs qidx (the plan rows of the batch)")
        rc = _lib.lib().mmssl_graph_sim_rows_f32(mask.handle, _ptr(Q), _ptr(qidx), B, _ptr(T), d, float(mask_value),
                                                 _ptr(out), ld, _ptr(part), _lib.stream_ptr())
        _lib.check(rc, "mmssl_graph_sim_rows_f32")
    else:
        rp, cols = (None, None) if mask is None else mask
        rc = _lib.lib().mmssl_sim_rows_f32(_ptr(Q), _ptr(qidx), B, _ptr(T), n, d, _ptr(rp), _ptr(cols), float(mask_value),
                                           _ptr(out), ld, _ptr(part), _lib.stream_ptr())
        _lib.check(rc, "mmssl_sim_rows_f32")
    inv = None
    if normalize:
        inv = torch.empty(B, dtype=torch.float32, device=dev)
        rc = _lib.lib().mmssl_rows_scale_parts_f32(_ptr(out), B, n, ld, _ptr(part), nparts, float(eps), _ptr(inv),
                                                   _lib.stream_ptr())
        _lib.check(rc, "mmssl_rows_scale_parts_f32")
    return (out if ld == n else out[:, :n]), inv


TOPK_MAX_COLS = 36864        # columns one mmssl_topk_rows_f32 launch ranks (144 KB of keys in LDS)
TOPK_MAX_K = 256


def _topk_launch(X, k, want_val):
    B, n = X.shape
    idx = torch.empty((B, k), dtype=torch.int64, device=X.device)
    val = torch.empty((B, k), dtype=torch.float32, device=X.device) if want_val else None
    rc = _lib.lib().mmssl_topk_rows_f32(_ptr(X), B, n, X.stride(0), int(k), _ptr(idx), _ptr(val), _lib.stream_ptr())
    _lib.check(rc, "mmssl_topk_rows_f32")
    return idx, val


def topk_rows(X, k, values=False):
    """Columns of the k largest entries of every row, descending value, ties by ascending column (the order
    heapq.nlargest gives the reference's evaluation, batch_test.py:21-36). X may be a row-pitched view. k <= 256.
    Rows wider than one launch ranks (36864 columns) go block by block: every column block's k winners, then one more
    launch over the winners (a winner of the whole row is a winner of its block; blocks and their winners are in
    ascending column order at equal value, so ties still break by ascending column)."""
    if not (X.is_cuda and X.dtype == torch.float32 and X.dim() == 2 and X.stride(1) == 1):
        raise _lib.MmsslError("topk_rows: fp32 [B, n] HIP tensor with unit column stride expected")
    if k < 1 or k > TOPK_MAX_K:
        raise _lib.MmsslError("topk_rows: 1 <= k <= %d expected, got %d" % (TOPK_MAX_K, k))
    B, n = X.shape
    if n <= TOPK_MAX_COLS:
        idx, val = _topk_launch(X, k, values)
        return (idx, val) if values else idx
    # equal blocks of at most TOPK_MAX_COLS columns, each a multiple of 4 wide (16-byte aligned starts)
    nblk = -(-n // TOPK_MAX_COLS)
    width = (-(-n // nblk) + 3) // 4 * 4
    # every block is at least n / nblk - 4 nblk > TOPK_MAX_K columns wide, so no block pads its winners with (-inf, -1)
    # slots that could tie with a later block's real -inf columns (the where() below only guards that invariant)
    if n - (nblk - 1) * width < k:
        raise _lib.MmsslError("topk_rows: column block narrower than k")
    cand_i, cand_v = [], []
    for c0 in range(0, n, width):
        i, v = _topk_launch(X[:, c0:min(n, c0 + width)], k, True)
        v = torch.where(i < 0, torch.full_like(v, float("-inf")), v)        # padding of a block narrower than k
        cand_i.append(torch.where(i < 0, i, i + c0))
        cand_v.append(v)
    ci, cv = torch.cat(cand_i, 1), torch.cat(cand_v, 1).contiguous()
    if ci.shape[1] > TOPK_MAX_COLS:
        raise _lib.MmsslError("topk_rows: %d columns x k = %d exceeds the merge launch" % (n, k))
    pos, val = _topk_launch(cv, k, True)
    idx = torch.gather(ci, 1, pos.clamp_min(0))
    idx = torch.where(pos < 0, pos, idx)
    return (idx, val) if values else idx


def rows_membership(rowptr, cols, rows, cand):
    """uint8 [B, K]: cand[b, k] in CSR row rows[b] (int32 rowptr / sorted cols on the device)."""
    B, K = cand.shape
    out = torch.empty((B, K), dtype=torch.uint8, device=cand.device)
    rc = _lib.lib().mmssl_rows_membership_u8(_ptr(rowptr), _ptr(cols), _ptr(rows), B, K, _ptr(cand.contiguous()), _ptr(out),
                                             _lib.stream_ptr())
    _lib.check(rc, "mmssl_rows_membership_u8")
    return out


def eval_accumulate(pos_rowptr, pos_cols, rows, cand, Ks, acc, ws=None):
    """acc [4, 8] float64 (precision, recall, ndcg, hit ratio @ Ks) += the sums over the batch's users, on the device
    (csrc/simtopk.hip eval_metrics_kernel; formulas of utility/batch_test.py:38-80). Returns the workspace for reuse."""
    B, K = cand.shape
    nb = _lib.lib().mmssl_eval_workspace_bytes(B)
    if ws is None or ws.numel() * 8 < nb:
        ws = torch.empty(nb // 8 + 2, dtype=torch.float64, device=cand.device)
    ks = (_ct.c_int * len(Ks))(*[int(k) for k in Ks])
    rc = _lib.lib().mmssl_eval_accumulate_f64(_ptr(pos_rowptr), _ptr(pos_cols), _ptr(rows), B, K, _ptr(cand.contiguous()), ks,
                                              len(Ks), _ptr(acc), _ptr(ws), ws.numel() * 8, _lib.stream_ptr())
    _lib.check(rc, "mmssl_eval_accumulate_f64")
    return ws


class _USim(torch.autograd.Function):
    """normalize((U[users] . I^T) * (1 - R[users]), dim=1) (main.py:283-298). Forward: the fused tile kernel of
    sim_rows (fp32 MFMA, gather of the batch rows, mask from the plan's device CSR, row norms from in-kernel
    partials). Backward: one kernel for the normalise/mask backward, then both products on this library's
    projection kernels: gU_b = gP . I (mmssl_linear_f32 against I^T) and gI = gP^T . U_b (mmssl_linear_wgrad_f32);
    S and gP use a 32-float row pitch so that the reduction dimension meets those kernels' alignment."""

    @staticmethod
    def forward(ctx, user_final, item_final, users, plan):
        user_final, item_final = _chk(user_final, "user_final"), _chk(item_final, "item_final")
        if plan.shape != (user_final.shape[0], item_final.shape[0]):
            raise _lib.MmsslError("usim: plan is %s, tables are [%d, d] / [%d, d]" % (
                plan.shape, user_final.shape[0], item_final.shape[0]))
        S, inv = sim_rows(user_final, item_final, qidx=users, mask=plan, mask_value=0.0, normalize=True, pitch_mult=32)
        ctx.save_for_backward(user_final, item_final, users, S, inv)
        ctx.plan = plan
        return S

    @staticmethod
    def backward(ctx, gS):
        user_final, item_final, users, S, inv = ctx.saved_tensors
        gS = _chk(gS, "gS")
        B, width = S.shape
        ld = S.stride(0)
        d = item_final.shape[1]
        dev = S.device
        gP = torch.zeros((B, ld), dtype=torch.float32, device=dev) if ld != width else torch.empty_like(S)
        rc = _lib.lib().mmssl_graph_rows_mask_normalize_bwd_ld_f32(ctx.plan.handle, _ptr(users), B, _ptr(S), ld, _ptr(gS),
                                                                   gS.stride(0), _ptr(inv), width, _NORM_EPS, _ptr(gP),
                                                                   ld, _lib.stream_ptr())
        _lib.check(rc, "mmssl_graph_rows_mask_normalize_bwd_ld_f32")
        gU = gI = None
        if ctx.needs_input_grad[0]:
            # I^T with the padded pitch [d, ld] (zero columns past n_items), then gU_b [B, d] = gP [B, ld] . (I^T)^T
            ItT = torch.empty((d, ld), dtype=torch.float32, device=dev)
            nbt = _lib.lib().mmssl_transpose_mask_workspace_bytes(ld, d)
            wst = torch.empty(max(nbt // 4, 4), dtype=torch.float32, device=dev)
            rc = _lib.lib().mmssl_transpose_mask_f32(_ptr(item_final), None, 1.0, width, d, ld, _ptr(ItT), None, _ptr(wst),
                                                     wst.numel() * 4, _lib.stream_ptr())
            _lib.check(rc, "mmssl_transpose_mask_f32")
            gUb = _linear_raw(gP, ItT, None, None, 1.0)
            gU = torch.zeros((user_final.shape[0], d), dtype=torch.float32, device=dev)
            gU.index_add_(0, users, gUb)
        if ctx.needs_input_grad[1]:
            # gI^T [d, ld] = U_b^T [d, B] . gP [B, ld]: the weight-gradient form (reduction over the batch)
            Ub = user_final.index_select(0, users)
            gIt = torch.empty((d, ld), dtype=torch.float32, device=dev)
            nb = _lib.lib().mmssl_linear_wgrad_workspace_bytes(B, ld, d)
            ws = torch.empty(max(nb // 4, 4), dtype=torch.float32, device=dev)
            rc = _lib.lib().mmssl_linear_wgrad_f32(_ptr(Ub), None, 1.0, _ptr(gP), B, ld, d, _ptr(gIt), None, _ptr(ws), nb,
                                                   _lib.stream_ptr())
            _lib.check(rc, "mmssl_linear_wgrad_f32")
            gI = gIt[:, :width].t().contiguous()
        return gU, gI, None, None


def usim(users, user_final, item_final, plan):
    """Trainer.u_sim_calculation: [B, n_items] masked, row-normalised scores of the batch users.
    `plan` is the GraphPlan of the user-item train graph (its sparsity pattern is the mask)."""
    if not hasattr(plan, "handle"):
        raise _lib.MmsslError("usim expects a GraphPlan")
    return _USim.apply(user_final, item_final, _idx(users, "users", user_final.device), plan)


def graph_rows_dense(plan, rows, value=1.0):
    """[len(rows), n_cols] dense rows of the plan's pattern (value where an edge exists, else 0): the
    reference's `torch.tensor(ui_graph_raw[users].todense()).cuda()` built on the device."""
    if not hasattr(plan, "handle"):
        raise _lib.MmsslError("graph_rows_dense expects a GraphPlan")
    dev = plan.device if hasattr(plan, "device") else torch.device("cuda")
    rows = _idx(rows, "rows", dev)
    out = torch.empty((rows.shape[0], plan.shape[1]), dtype=torch.float32, device=rows.device)
    rc = _lib.lib().mmssl_graph_rows_dense_f32(plan.handle, _ptr(rows), rows.shape[0], float(value), _ptr(out),
                                               plan.shape[1], _lib.stream_ptr())
    _lib.check(rc, "mmssl_graph_rows_dense_f32")
    return out


# ---------------------------------------------------------------------------------------
# all batch losses in one autograd node          main.py:368-371, 411-412, 499-511
# ---------------------------------------------------------------------------------------
class _BatchLosses(torch.autograd.Function):
    """(mf, emb, cl_img, cl_txt) from the full tables and the batch indices: fused-gather BPR and
    two fused-gather InfoNCE calls that share the user table. The backward zero-fills each table
    gradient ONCE and all three kernels scatter-add into it (instead of three dense gradients
    summed by autograd)."""

    @staticmethod
    def forward(ctx, ua, ia, img_uid, txt_uid, users, pos, neg, decay, batch_size, tau, hot=None, eager_w=None,
                tail=None):
        ua, ia = _chk(ua, "ua"), _chk(ia, "ia")
        img_uid, txt_uid = _chk(img_uid, "img_uid"), _chk(txt_uid, "txt_uid")
        B, d = users.shape[0], ua.shape[1]
        dev = ua.device
        out = torch.empty(5, dtype=torch.float32, device=dev)     # mf, emb, reg(=0), cl_img, cl_txt
        nb = _lib.lib().mmssl_bpr_workspace_bytes(B)
        wsb = torch.empty(nb // 4, dtype=torch.float32, device=dev)
        need_grad = any(ctx.needs_input_grad[:4])
        if eager_w is not None and tail is not None and need_grad:
            return _BatchLosses._forward_eager(ctx, ua, ia, img_uid, txt_uid, users, pos, neg, decay, batch_size, tau,
                                               eager_w, tail, out, wsb, nb, hot)
        ctx.eager = None
        # the table gradients the backward scatter-adds into are allocated and zero-filled here
        g_ua = torch.zeros_like(ua) if need_grad else None
        g_ia = torch.zeros_like(ia) if need_grad else None
        rc = _lib.lib().mmssl_bpr_fwd_f32(_ptr(ua), _ptr(ia), None, _ptr(users), _ptr(pos), _ptr(neg), B, d,
                                          float(decay), int(batch_size), _ptr(out), _ptr(wsb), nb, _lib.stream_ptr())
        _lib.check(rc, "mmssl_bpr_fwd_f32")
        # both InfoNCE problems (image / text view vs the same user table) in ONE set of launches,
        # losses written straight into out[3], out[4]
        nbw = _lib.lib().mmssl_infonce_multi_workspace_bytes(2, B, d)
        if nbw == 0:
            raise _lib.MmsslError("infonce: unsupported shape n=%d d=%d" % (B, d))
        ws1 = torch.empty(nbw // 4, dtype=torch.float32, device=dev)
        z1s = (_ct.c_void_p * 2)(img_uid.data_ptr(), txt_uid.data_ptr())
        rc = _lib.lib().mmssl_infonce_multi_fwd_f32(z1s, _ptr(ua), _ptr(users), 2, B, d, float(tau), _ptr(out[3:5]),
                                                    _ptr(ws1), nbw, _lib.stream_ptr())
        _lib.check(rc, "mmssl_infonce_multi_fwd_f32")
        ctx.save_for_backward(ua, ia, users, pos, neg, ws1)
        ctx.gbuf = (g_ua, g_ia)
        ctx.cfg = (B, d, float(decay), int(batch_size), float(tau), img_uid.shape, txt_uid.shape)
        return out

    @staticmethod
    def _forward_eager(ctx, ua, ia, img_uid, txt_uid, users, pos, neg, decay, batch_size, tau, w, tail, out, wsb, nb, hot):
        """The caller PROMISES that the result is backpropagated with exactly the gradient `w` (a persistent [5]
        tensor: HotPathStep's loss weights, total = w . terms + c * extra backpropagated with 1). The gradients of
        the loss terms are then known before the loss scalars are, and the whole loss section becomes ONE chain on the
        current stream, no fork / join (a cross-queue edge of a replayed hipGraph costs 10-15 us). d <= 64 (four launches
        behind the zero fill of the gradients, which the step's forward already did on an idle stream): InfoNCE prep +
        BPR rows, pair tiles, backward pair tiles (per-row terms on the fly), backward finish + InfoNCE losses + BPR loss
        + loss assembly + counter ticks. Wider rows: prep, pair tiles, row terms (the last block also reduces the two
        losses), backward pair tiles, backward finish, and mmssl_bpr_step_f32 as a launch of its own.
        tail = (extra, c, total, ticks): see mmssl_bpr_step_f32. backward() returns the stored gradients.
        `hot` (hotnode.HotCtx of the step, may be None): hands over the zero-filled buffer the forward prepared on an idle
        stream (hot.prefill_buf) and the forward's unreduced regulariser partials (hot.ss_parts)."""
        B, d = users.shape[0], ua.shape[1]
        dev = ua.device
        if w.dtype != torch.float32 or w.numel() != 5 or w.device != dev:
            raise _lib.MmsslError("batch_losses: eager_w must be a [5] fp32 tensor on the tables' device")
        extra, c, total, ticks = tail
        # gradients of the modal views only where they are wanted (the cached all-zero views of empty modal graphs need
        # none: no buffer, no fill, no store)
        need_im, need_tx = bool(ctx.needs_input_grad[2]), bool(ctx.needs_input_grad[3])
        n_ua, n_ia = ua.numel(), ia.numel()
        n_im, n_tx = (img_uid.numel() if need_im else 0), (txt_uid.numel() if need_tx else 0)
        gbuf = None
        if hot is not None:
            gbuf, hot.prefill_buf = hot.prefill_buf, None          # zero-filled by the forward, if enabled
        if gbuf is None or gbuf.numel() != n_ua + n_ia + n_im + n_tx + 4 or gbuf.device != dev:
            gbuf = torch.zeros(n_ua + n_ia + n_im + n_tx + 4, dtype=torch.float32, device=dev)
        g_ua = gbuf[:n_ua].view_as(ua)
        g_ia = gbuf[n_ua:n_ua + n_ia].view_as(ia)
        g_img = gbuf[n_ua + n_ia:n_ua + n_ia + n_im].view_as(img_uid) if need_im else None
        g_txt = gbuf[n_ua + n_ia + n_im:n_ua + n_ia + n_im + n_tx].view_as(txt_uid) if need_tx else None
        tickets = gbuf[n_ua + n_ia + n_im + n_tx:]              # three zeroed ints (InfoNCE x2, BPR)
        nbw = _lib.lib().mmssl_infonce_multi_workspace_bytes(2, B, d)
        if nbw == 0:
            raise _lib.MmsslError("infonce: unsupported shape n=%d d=%d" % (B, d))
        ws1 = torch.empty(nbw // 4, dtype=torch.float32, device=dev)
        z1s = (_ct.c_void_p * 2)(img_uid.data_ptr(), txt_uid.data_ptr())
        gz1s = (_ct.c_void_p * 2)(_ptr(g_img), _ptr(g_txt))
        f32s, u64s = ticks if ticks else ((), ())
        fa = (_ct.c_void_p * max(len(f32s), 1))(*[int(x) for x in f32s])
        ka = (_ct.c_void_p * max(len(u64s), 1))(*[int(x) for x in u64s])
        xparts, n_xparts = None, 0
        last = hot.ss_parts if hot is not None else None
        if last is not None and extra is not None and last[0].data_ptr() == extra.data_ptr():
            xparts, n_xparts = last[1], last[1].numel()       # `extra` is a forward's unreduced regulariser sum
            hot.ss_parts = None
        elif last is not None:
            raise _lib.MmsslError("batch_losses: the forward left an unreduced regulariser sum that this tail does not consume")
        if d <= 64:
            # FOUR launches: InfoNCE prep (+ the BPR tail's ROWS part as guest blocks: it depends on nothing there and
            # the short launch leaves most of the chip free), forward pair tiles, backward pair tiles (exactly one round
            # of resident blocks without guests; the per-row terms come from the forward's partial denominators on the
            # fly), backward finish (+ one guest block: the two InfoNCE losses from the per-tile partials, then the
            # tail's ASSEMBLY part: BPR loss, loss assembly, counter ticks)
            rc = _lib.lib().mmssl_infonce_multi_fwd_ticket_bpr_f32(
                z1s, _ptr(ua), _ptr(users), 2, B, d, float(tau), _ptr(out[3:5]), _ptr(ws1), nbw, _ptr(tickets),
                _ptr(ua), _ptr(ia), _ptr(users), _ptr(pos), _ptr(neg), B, float(decay), int(batch_size), _ptr(w[0:1]),
                _ptr(w[1:2]), _ptr(g_ua), _ptr(g_ia), _ptr(wsb), nb, _lib.stream_ptr())
            _lib.check(rc, "mmssl_infonce_multi_fwd_ticket_bpr_f32")
            rc = _lib.lib().mmssl_infonce_multi_bwd_phase_f32(_ptr(users), 2, B, d, float(tau), _ptr(w[3:5]), gz1s,
                                                              _ptr(g_ua), _ptr(ws1), ws1.numel() * 4, 1, _lib.stream_ptr())
            _lib.check(rc, "mmssl_infonce_multi_bwd_phase_f32")
            rc = _lib.lib().mmssl_infonce_multi_bwd_finish_bpr_f32(
                _ptr(users), 2, B, d, float(tau), _ptr(w[3:5]), _ptr(out[3:5]), gz1s, _ptr(g_ua), _ptr(ws1), ws1.numel() * 4,
                _ptr(ua), _ptr(ia), _ptr(users), _ptr(pos), _ptr(neg), B, float(decay), int(batch_size), _ptr(w[0:1]),
                _ptr(w[1:2]), _ptr(g_ua), _ptr(g_ia), _ptr(out), _ptr(w), 5, _ptr(extra), float(c), _ptr(total), fa,
                len(f32s), ka, len(u64s), _ptr(wsb), nb, _ptr(xparts), n_xparts, _lib.stream_ptr())
            _lib.check(rc, "mmssl_infonce_multi_bwd_finish_bpr_f32")
            ctx.eager = (g_ua, g_ia, g_img, g_txt)
            return out
        rc = _lib.lib().mmssl_infonce_multi_fwd_ticket_f32(z1s, _ptr(ua), _ptr(users), 2, B, d, float(tau), _ptr(out[3:5]),
                                                           _ptr(ws1), nbw, _ptr(tickets), _lib.stream_ptr())
        _lib.check(rc, "mmssl_infonce_multi_fwd_ticket_f32")
        rc = _lib.lib().mmssl_infonce_multi_bwd_phase_f32(_ptr(users), 2, B, d, float(tau), _ptr(w[3:5]), gz1s, _ptr(g_ua),
                                                          _ptr(ws1), ws1.numel() * 4, 3, _lib.stream_ptr())
        _lib.check(rc, "mmssl_infonce_multi_bwd_phase_f32")
        rc = _lib.lib().mmssl_bpr_step_f32(_ptr(ua), _ptr(ia), _ptr(users), _ptr(pos), _ptr(neg), B, d, float(decay),
                                           int(batch_size), _ptr(w[0:1]), _ptr(w[1:2]), _ptr(g_ua), _ptr(g_ia), _ptr(out),
                                           _ptr(w), 5, _ptr(extra), float(c), _ptr(total), fa, len(f32s), ka, len(u64s),
                                           _ptr(wsb), nb, _ptr(tickets[2:]), _ptr(xparts), n_xparts, _lib.stream_ptr())
        _lib.check(rc, "mmssl_bpr_step_f32")
        ctx.eager = (g_ua, g_ia, g_img, g_txt)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.eager is not None:
            g_ua, g_ia, g_img, g_txt = ctx.eager
            ctx.eager = None
            return g_ua, g_ia, g_img, g_txt, None, None, None, None, None, None, None, None, None
        ua, ia, users, pos, neg, ws1 = ctx.saved_tensors
        B, d, decay, batch_size, tau, s_img, s_txt = ctx.cfg
        g = g.contiguous().to(torch.float32)
        g_ua, g_ia = ctx.gbuf
        ctx.gbuf = None
        if g_ua is None:                     # second backward through a retained graph
            g_ua, g_ia = torch.zeros_like(ua), torch.zeros_like(ia)
        dev = ua.device
        g_img = torch.zeros(s_img, dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
        g_txt = torch.zeros(s_txt, dtype=torch.float32, device=dev) if ctx.needs_input_grad[3] else None
        gz1s = (_ct.c_void_p * 2)(_ptr(g_img), _ptr(g_txt))
        # BPR backward (scatter-add into g_ua / g_ia), the InfoNCE pair tiles (workspace only), then the InfoNCE finish
        # (scatter-adds into g_ua)
        rc = _lib.lib().mmssl_bpr_bwd_f32(_ptr(ua), _ptr(ia), None, _ptr(users), _ptr(pos), _ptr(neg), B, d,
                                          decay, batch_size, _ptr(g[0:1]), _ptr(g[1:2]), _ptr(g_ua), _ptr(g_ia),
                                          None, _lib.stream_ptr())
        _lib.check(rc, "mmssl_bpr_bwd_f32")
        call = _lib.lib().mmssl_infonce_multi_bwd_phase_f32
        rc = call(_ptr(users), 2, B, d, tau, _ptr(g[3:5]), gz1s, _ptr(g_ua), _ptr(ws1), ws1.numel() * 4, 1,
                  _lib.stream_ptr())
        _lib.check(rc, "mmssl_infonce_multi_bwd_phase_f32")
        rc = call(_ptr(users), 2, B, d, tau, _ptr(g[3:5]), gz1s, _ptr(g_ua), _ptr(ws1), ws1.numel() * 4, 2,
                  _lib.stream_ptr())
        _lib.check(rc, "mmssl_infonce_multi_bwd_phase_f32")
        return g_ua, g_ia, g_img, g_txt, None, None, None, None, None, None, None, None, None


def batch_losses_vec(ua, ia, img_uid, txt_uid, users, pos, neg, decay, batch_size, tau, hot=None, eager_w=None,
                     tail=None):
    """[mf_loss, emb_loss, 0, cl_img, cl_txt] as ONE tensor (see _BatchLosses / loss_assemble).
    eager_w + tail=(extra, c, total, ticks): the caller's promise that the result is backpropagated with exactly the
    [5] gradient eager_w, total = eager_w . terms + c * extra being written to `total` by the same launches
    (see _BatchLosses._forward_eager); `hot` = the step's hotnode.HotCtx (hand-offs from the forward node)."""
    dev = ua.device
    if eager_w is not None and tail is None:
        eager_w = None
    return _BatchLosses.apply(ua, ia, img_uid, txt_uid, _idx(users, "users", dev), _idx(pos, "pos", dev),
                              _idx(neg, "neg", dev), decay, batch_size, tau, hot, eager_w, tail)


def batch_losses(ua, ia, img_uid, txt_uid, users, pos, neg, decay, batch_size, tau):
    """Returns (mf_loss, emb_loss, cl_img, cl_txt) — see _BatchLosses."""
    out = batch_losses_vec(ua, ia, img_uid, txt_uid, users, pos, neg, decay, batch_size, tau)
    return out[0], out[1], out[3], out[4]


class _LossAssemble(torch.autograd.Function):
    """total = sum_k w[k] * terms[k] + c * extra   (main.py:420) in one launch, written into `out` when
    given (a persistent buffer: no copy of the step's loss afterwards); the backward is one launch too,
    instead of the ~20 scalar autograd kernels of the op-by-op expression. With `unit_grad` the caller
    PROMISES to backpropagate a gradient of exactly 1 (HotPathStep passes its persistent ones tensor): the
    gradients are then the constants w and c themselves and the backward launches nothing."""

    @staticmethod
    def forward(ctx, terms, w, extra, c, out, unit_grad, ticks=None):
        if out is not None:
            # a fresh tensor object over `out`'s memory: the result carries the autograd history, the
            # caller's buffer stays a plain leaf that can be passed again next step
            total = torch.empty(0, dtype=torch.float32, device=out.device).set_(
                out.untyped_storage(), out.storage_offset(), torch.Size(()), ())
        else:
            total = torch.empty((), dtype=torch.float32, device=terms.device)
        if ticks:
            f32s, u64s = ticks
            fa = (_ct.c_void_p * max(len(f32s), 1))(*[int(x) for x in f32s])
            ua = (_ct.c_void_p * max(len(u64s), 1))(*[int(x) for x in u64s])
            rc = _lib.lib().mmssl_loss_assemble_tick_f32(_ptr(terms), _ptr(w), terms.numel(), _ptr(extra), float(c),
                                                         _ptr(total), fa, len(f32s), ua, len(u64s), _lib.stream_ptr())
            _lib.check(rc, "mmssl_loss_assemble_tick_f32")
        else:
            rc = _lib.lib().mmssl_loss_assemble_f32(_ptr(terms), _ptr(w), terms.numel(), _ptr(extra), float(c),
                                                    _ptr(total), _lib.stream_ptr())
            _lib.check(rc, "mmssl_loss_assemble_f32")
        ctx.save_for_backward(w)
        ctx.c = float(c)
        ctx.has_extra = extra is not None
        ctx.unit = unit_grad if (unit_grad is None or isinstance(unit_grad, torch.Tensor)) else None
        return total

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        if ctx.unit is not None:             # promised d(total) == 1: gradients are the constants themselves
            return w, None, (ctx.unit if ctx.has_extra else None), None, None, None, None
        g = g.contiguous().to(torch.float32)
        gt = torch.empty_like(w)
        ge = torch.empty((), dtype=torch.float32, device=w.device) if ctx.has_extra else None
        rc = _lib.lib().mmssl_loss_assemble_bwd_f32(_ptr(g), _ptr(w), w.numel(), ctx.c, _ptr(gt), _ptr(ge),
                                                    _lib.stream_ptr())
        _lib.check(rc, "mmssl_loss_assemble_bwd_f32")
        return gt, None, ge, None, None, None, None


def loss_assemble(terms, w, extra=None, c=0.0, out=None, unit_grad_c=None, ticks=None):
    """`unit_grad_c`: a persistent 0-dim tensor holding the value `c`; passing it is the promise that the
    result is backpropagated with a gradient of exactly 1 (see _LossAssemble). `ticks` = (float counter
    addresses, uint64 counter addresses) advanced by one in the same launch (see external_ticks)."""
    if out is not None and (out.dtype != torch.float32 or out.numel() != 1 or not out.is_cuda or out.requires_grad):
        raise _lib.MmsslError("loss_assemble: `out` must be a one-element fp32 HIP tensor that needs no gradient")
    return _LossAssemble.apply(_chk(terms, "terms"), _chk(w, "w"), extra, c, out, unit_grad_c, ticks)


class _ZeroGradAnchor(torch.autograd.Function):
    """Identity on `x` that also makes the result depend on `w` with an exactly-zero gradient
    (used when a provably-zero branch of the reference graph is skipped, so that the optimiser
    still sees a zero — not a missing — gradient for `w`, like the reference's autograd).
    With a context whose `lazy_anchors` is set (hotnode.HotCtx of a step object) the backward launches nothing: the step
    assigns a persistent all-zero `.grad` afterwards (HotCtx.assign_anchored_zero_grads)."""

    @staticmethod
    def forward(ctx, x, w, hot):
        ctx.wshape, ctx.wdev = w.shape, w.device
        ctx.lazy = hot is not None and hot.lazy_anchors
        if ctx.lazy:
            hot.anchored.append(w)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if ctx.lazy:
            return g, None, None
        return g, torch.zeros(ctx.wshape, dtype=torch.float32, device=ctx.wdev), None


def zero_grad_anchor(x, w, hot=None):
    return _ZeroGradAnchor.apply(x, w, hot)


# ---------------------------------------------------------------------------------------
# layer mean + modality fusion kernels (Models.py:213-218) and the projection's mask epilogue
# ---------------------------------------------------------------------------------------
def _combine_fwd(layers, inv, A, B, r, part):
    out = torch.empty_like(A)
    arr = (_ct.c_void_p * len(layers))(*[t.data_ptr() for t in layers])
    rc = _lib.lib().mmssl_layer_combine_f32(arr, len(layers), float(inv), _ptr(A), _ptr(B), float(r), A.shape[0],
                                            A.shape[1], _NORM_EPS, _ptr(out), _ptr(part), _lib.stream_ptr())
    _lib.check(rc, "mmssl_layer_combine_f32")
    return out


def _combine_bwd(A, B, G, r, inv, c_dev, c_scale, want_gL):
    gA, gB = torch.empty_like(A), torch.empty_like(B)
    gL = torch.empty_like(A) if want_gL else None
    rc = _lib.lib().mmssl_layer_combine_bwd_f32(_ptr(A), _ptr(B), _ptr(G), float(r), float(inv), _ptr(c_dev),
                                                float(c_scale), A.shape[0], A.shape[1], _NORM_EPS, _ptr(gA), _ptr(gB),
                                                _ptr(gL), _lib.stream_ptr())
    _lib.check(rc, "mmssl_layer_combine_bwd_f32")
    return gA, gB, gL


def spmm_mask_raw(plan, transpose, X, keep, dm, scale):
    """Y = keep ? (op(A) . X) * scale : 0 with Y [rows, d] packing d / dm modalities side by side and keep the uint8
    [d / dm, rows, dm] mask of proj_forward: the projection's dropout backward fused into the SpMM store."""
    rows = plan.shape[1] if transpose else plan.shape[0]
    cols = plan.shape[0] if transpose else plan.shape[1]
    if X.dim() != 2 or X.shape[0] != cols:
        raise _lib.MmsslError("spmm: X has shape %s, expected [%d, d]" % (tuple(X.shape), cols))
    d = X.shape[1]
    if keep.dtype != torch.uint8 or keep.numel() != rows * d or not keep.is_contiguous():
        raise _lib.MmsslError("spmm_mask: keep must be a contiguous uint8 [d / dm, rows, dm] tensor")
    if STATS["enabled"]:
        _count_spmm(plan, rows, d)
    Y = torch.empty((rows, d), dtype=torch.float32, device=X.device)
    ws = plan.workspace(transpose, d)
    rc = _lib.lib().mmssl_spmm_mask_f32(plan.handle, int(transpose), _ptr(X), d, _ptr(Y), _ptr(keep), int(dm), float(scale),
                                        _ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "mmssl_spmm_mask_f32")
    return Y


def mask_packed(G, keep, dm, scale):
    """keep ? G * scale : 0 for packed modal rows G [rows, nm * dm] and the uint8 [nm, rows, dm] masks of proj_forward
    (in place: the dropout backward of the row-sharded step, after its reduce-scatter)."""
    rows, d = G.shape
    if keep.dtype != torch.uint8 or keep.numel() != rows * d or not keep.is_contiguous() or not G.is_contiguous():
        raise _lib.MmsslError("mask_packed: contiguous G [rows, nm * dm] and uint8 keep [nm, rows, dm] expected")
    rc = _lib.lib().mmssl_mask_packed_f32(_ptr(G), _ptr(keep), float(scale), rows, d // dm, int(dm), _ptr(G),
                                          _lib.stream_ptr())
    _lib.check(rc, "mmssl_mask_packed_f32")
    return G


def fuse_blocks(rows, d, nm):
    """Number of per-block |Mod|^2 partial sums fuse_fwd writes for a side of `rows` rows."""
    return int(_lib.lib().mmssl_fuse_blocks(int(rows), int(d), int(nm)))


def fuse_fwd(sides, inv, nm, r):
    """Layer mean + modality fusion over packed modal features, one launch for all `sides` (1 or 2: user tables, item
    tables). sides = [(layers, Mod [rows, nm * d], part or None), ...]: out = inv * sum(layers) + r * sum_m
    normalize(Mod[:, m-th slice]); part (fuse_blocks(rows, d, nm) floats) receives the partial sums of |Mod|^2.
    Returns the list of outputs."""
    n = len(sides)
    outs = [torch.empty_like(sd[0][0]) for sd in sides]
    d = outs[0].shape[1]
    lay = [(_ct.c_void_p * len(sd[0]))(*[t.data_ptr() for t in sd[0]]) for sd in sides]
    lay_arr = (_ct.c_void_p * n)(*[_ct.cast(a, _ct.c_void_p).value for a in lay])
    mods = (_ct.c_void_p * n)(*[sd[1].data_ptr() for sd in sides])
    rows = (_ct.c_int64 * n)(*[o.shape[0] for o in outs])
    oarr = (_ct.c_void_p * n)(*[o.data_ptr() for o in outs])
    parr = (_ct.c_void_p * n)(*[None if sd[2] is None else sd[2].data_ptr() for sd in sides])
    rc = _lib.lib().mmssl_fuse_fwd_f32(n, lay_arr, len(sides[0][0]), float(inv), mods, int(nm), float(r), rows, d, _NORM_EPS,
                                       oarr, parr, _lib.stream_ptr())
    _lib.check(rc, "mmssl_fuse_fwd_f32")
    return outs


def fuse_fwd_rows(sides, inv, nm, r, lo=None):
    """fuse_fwd for a LIST of rows per side: sides = [(layers, Mod [rows, nm * d], idx int64 [n]), ...]. Returns full-size
    output tables of which ONLY the listed rows are defined (bit for bit fuse_fwd's values there). `lo` (one int per
    side): the lists hold GLOBAL row ids of row-sharded tables whose local rows start at lo[k]; rows of other ranks are
    skipped (mmssl_fuse_fwd_owned_rows_f32)."""
    n = len(sides)
    outs = [torch.empty_like(sd[0][0]) for sd in sides]
    d = outs[0].shape[1]
    lay = [(_ct.c_void_p * len(sd[0]))(*[t.data_ptr() for t in sd[0]]) for sd in sides]
    lay_arr = (_ct.c_void_p * n)(*[_ct.cast(a, _ct.c_void_p).value for a in lay])
    mods = (_ct.c_void_p * n)(*[sd[1].data_ptr() for sd in sides])
    idxs = (_ct.c_void_p * n)(*[sd[2].data_ptr() for sd in sides])
    nidx = (_ct.c_int64 * n)(*[sd[2].numel() for sd in sides])
    oarr = (_ct.c_void_p * n)(*[o.data_ptr() for o in outs])
    for sd in sides:
        if sd[2].dtype != torch.int64 or not sd[2].is_contiguous():
            raise _lib.MmsslError("fuse_fwd_rows: contiguous int64 row lists expected")
    if lo is not None:
        los = (_ct.c_int64 * n)(*[int(x) for x in lo])
        nloc = (_ct.c_int64 * n)(*[sd[0][0].shape[0] for sd in sides])
        rc = _lib.lib().mmssl_fuse_fwd_owned_rows_f32(n, lay_arr, len(sides[0][0]), float(inv), mods, int(nm), float(r), idxs,
                                                      nidx, los, nloc, d, _NORM_EPS, oarr, _lib.stream_ptr())
        _lib.check(rc, "mmssl_fuse_fwd_owned_rows_f32")
        return outs
    rc = _lib.lib().mmssl_fuse_fwd_rows_f32(n, lay_arr, len(sides[0][0]), float(inv), mods, int(nm), float(r), idxs, nidx, d,
                                            _NORM_EPS, oarr, _lib.stream_ptr())
    _lib.check(rc, "mmssl_fuse_fwd_rows_f32")
    return outs


def loss_add_partials(part, c, total, sum_out=None):
    """total += c * sum(part) (one launch); sum_out receives the sum."""
    rc = _lib.lib().mmssl_loss_add_partials_f32(_ptr(part), part.numel(), float(c), _ptr(total), _ptr(sum_out),
                                                _lib.stream_ptr())
    _lib.check(rc, "mmssl_loss_add_partials_f32")


def fuse_bwd(sides, nm, r, inv, c_dev, c_scale, sumsq_part=None):
    """The backward of fuse_fwd, one launch for all sides. sides = [(Mod, G, Gx or None, want_gL), ...]; returns
    [(gMod [rows, nm * d], gL or None), ...]: gMod = r * normalize_bwd(Mod_m, G) + (c_scale * c_dev) * Mod_m (+ Gx),
    gL = inv * G. `sumsq_part`: one float tensor of fuse_blocks(rows, d, nm) entries per side, receives the per-block
    sums of |Mod|^2 (the norms the kernel computes anyway)."""
    n = len(sides)
    gMods = [torch.empty_like(sd[0]) for sd in sides]
    gLs = [torch.empty_like(sd[1]) if sd[3] else None for sd in sides]
    d = sides[0][1].shape[1]
    arr = lambda ts: (_ct.c_void_p * n)(*[None if t is None else t.data_ptr() for t in ts])       # noqa: E731
    rows = (_ct.c_int64 * n)(*[sd[1].shape[0] for sd in sides])
    rc = _lib.lib().mmssl_fuse_bwd_f32(n, arr([sd[0] for sd in sides]), int(nm), arr([sd[1] for sd in sides]),
                                       arr([sd[2] for sd in sides]), float(r), float(inv), _ptr(c_dev), float(c_scale), rows,
                                       d, _NORM_EPS, arr(gMods), arr(gLs), arr(sumsq_part) if sumsq_part else None,
                                       _lib.stream_ptr())
    _lib.check(rc, "mmssl_fuse_bwd_f32")
    return list(zip(gMods, gLs))


# ---------------------------------------------------------------------------------------
# Baseline models (csrc/baselines.hip): kNN-list item-graph product, NGCF layer tail
# ---------------------------------------------------------------------------------------
class _EllSpmm(torch.autograd.Function):
    """y[i] = sum_j w[i, j] * h[idx[i, j]] (LATTICE Models.py:103-104 / MICRO Models.py:112-118 on neighbour LISTS);
    backward: the transposed product for h (fp32 atomics) and one dot product per stored entry for the learned weights."""

    @staticmethod
    def forward(ctx, idx, w, h):
        w, h = _chk(w, "w"), _chk(h, "h")
        idx = _idx(idx, "idx", h.device)
        if idx.dim() != 2 or tuple(w.shape) != tuple(idx.shape):
            raise _lib.MmsslError("ell_spmm: idx / w must be [rows, k]")
        rows, k = idx.shape
        y = torch.empty((rows, h.shape[1]), dtype=torch.float32, device=h.device)
        rc = _lib.lib().mmssl_ell_spmm_f32(_ptr(idx), _ptr(w), rows, k, _ptr(h), h.shape[1], _ptr(y), _lib.stream_ptr())
        _lib.check(rc, "mmssl_ell_spmm_f32")
        ctx.save_for_backward(idx, w, h)
        return y

    @staticmethod
    def backward(ctx, gy):
        idx, w, h = ctx.saved_tensors
        gy = _chk(gy, "gy")
        rows, k = idx.shape
        gw = torch.empty_like(w) if ctx.needs_input_grad[1] else None
        gh = torch.zeros_like(h) if ctx.needs_input_grad[2] else None
        if gw is not None or gh is not None:
            rc = _lib.lib().mmssl_ell_spmm_bwd_f32(_ptr(idx), _ptr(w), rows, k, _ptr(h), h.shape[1], _ptr(gy), _ptr(gw),
                                                   _ptr(gh), _lib.stream_ptr())
            _lib.check(rc, "mmssl_ell_spmm_bwd_f32")
        return None, gw, gh


def ell_spmm(idx, w, h):
    return _EllSpmm.apply(idx, w, h)


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _chk(a, "a"), _chk(b, "b")
        out = torch.empty_like(a)
        _lib.check(_lib.lib().mmssl_mul_f32(_ptr(a), _ptr(b), a.numel(), _ptr(out), _lib.stream_ptr()), "mmssl_mul_f32")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _chk(g, "g")
        ga, gb = torch.empty_like(a), torch.empty_like(b)
        rc = _lib.lib().mmssl_mul_bwd_f32(_ptr(a), _ptr(b), _ptr(g), a.numel(), _ptr(ga), _ptr(gb), _lib.stream_ptr())
        _lib.check(rc, "mmssl_mul_bwd_f32")
        return ga, gb


def mul(a, b):
    """Elementwise product of two [rows, d] fp32 tensors (NGCF's ego * side) with its two gradients in one launch."""
    return _Mul.apply(a, b)


class _NGCFCombine(torch.autograd.Function):
    """(ego', norm) = (dropout(leaky_relu(G) + leaky_relu(B)), normalize(ego')) in one launch; one launch backward."""

    @staticmethod
    def forward(ctx, G, B, keep, scale):
        G, B = _chk(G, "G"), _chk(B, "B")
        ego, norm = torch.empty_like(G), torch.empty_like(G)
        rc = _lib.lib().mmssl_ngcf_combine_f32(_ptr(G), _ptr(B), _ptr(keep), float(scale), G.shape[0], G.shape[1], _NORM_EPS,
                                               _ptr(ego), _ptr(norm), _lib.stream_ptr())
        _lib.check(rc, "mmssl_ngcf_combine_f32")
        ctx.save_for_backward(G, B, keep, ego)
        ctx.scale = float(scale)
        ctx.set_materialize_grads(False)
        return ego, norm

    @staticmethod
    def backward(ctx, g_ego, g_norm):
        G, B, keep, ego = ctx.saved_tensors
        if g_ego is None and g_norm is None:
            return None, None, None, None
        g_ego = _chk(g_ego, "g_ego") if g_ego is not None else None
        g_norm = _chk(g_norm, "g_norm") if g_norm is not None else None
        gG, gB = torch.empty_like(G), torch.empty_like(B)
        rc = _lib.lib().mmssl_ngcf_combine_bwd_f32(_ptr(G), _ptr(B), _ptr(keep), ctx.scale, _ptr(ego), _ptr(g_ego),
                                                   _ptr(g_norm), G.shape[0], G.shape[1], _NORM_EPS, _ptr(gG), _ptr(gB),
                                                   _lib.stream_ptr())
        _lib.check(rc, "mmssl_ngcf_combine_bwd_f32")
        return gG, gB, None, None


def ngcf_combine(G, B, keep=None, scale=1.0):
    return _NGCFCombine.apply(G, B, keep, scale)
