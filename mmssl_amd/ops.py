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
STATS = {"enabled": False, "spmm_launches": 0, "edge_layers": 0, "spmm_bytes": 0}
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
def _spmm_raw(plan, transpose, X, epilogue, Z=None, alpha=0.0, S=None):
    rows = plan.shape[1] if transpose else plan.shape[0]
    cols = plan.shape[0] if transpose else plan.shape[1]
    if X.dim() != 2 or X.shape[0] != cols:
        raise _lib.MmsslError("spmm: X has shape %s, expected [%d, d]" % (tuple(X.shape), cols))
    d = X.shape[1]
    if STATS["enabled"]:
        STATS["spmm_launches"] += 1
        STATS["edge_layers"] += plan.nnz
        STATS["spmm_bytes"] += plan.nnz * (8 + 4 * d) + rows * 4 * d + (rows + 1) * 4
    Y = torch.empty((rows, d), dtype=torch.float32, device=X.device)
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
# OPT-IN split-precision projection (MMSSL_GEMM_SPLIT=1; default OFF = exact fp32 MFMA arithmetic).
# The constant feature matrix is split ONCE into bf16 (hi, lo) pairs (same bytes as fp32), forward and
# transposed-for-wgrad; W and gY are split per call. See csrc/linear.hip (gemm_split_kernel) and DESIGN.md.
_SPLIT = {}


def split_projection_enabled():
    return _os.environ.get("MMSSL_GEMM_SPLIT", "0") == "1"


def _bf16_pair(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi.contiguous(), lo.contiguous()


def register_split_features(F_):
    """Declare `F_` a CONSTANT feature matrix and build its split copies once: (F_hi, F_lo [M, K]) and the
    transposed pair padded along M to a multiple of 128 (the wgrad reduction runs over M). Only registered
    tensors take the split path (anything else - activations, test inputs - stays on the exact fp32 kernels)."""
    key = (F_.data_ptr(), tuple(F_.shape))
    if key not in _SPLIT:
        M, K = F_.shape
        Mp = (M + 127) // 128 * 128
        FT = torch.zeros((K, Mp), dtype=torch.float32, device=F_.device)
        FT[:, :M] = F_.t()
        _SPLIT[key] = (_bf16_pair(F_), _bf16_pair(FT), Mp, F_)          # keeps F_ alive: the key stays valid
        del FT
    return _SPLIT[key]


def _split_features(F_):
    return _SPLIT.get((F_.data_ptr(), tuple(F_.shape)))


def _use_split(F_):
    K = F_.shape[1]
    return split_projection_enabled() and K % 32 == 0 and K >= 96 and _split_features(F_) is not None


def _split_call(Ah, Al, Bh, Bl, b, keep, scale, M, K, N):
    Y = torch.empty((M, N), dtype=torch.float32, device=Ah.device)
    nb = _lib.lib().mmssl_linear_split_workspace_bytes(M, K, N)
    ws = torch.empty(max(nb // 4, 4), dtype=torch.float32, device=Ah.device)
    rc = _lib.lib().mmssl_linear_split_f32(_ptr(Ah), _ptr(Al), _ptr(Bh), _ptr(Bl), _ptr(b), _ptr(keep), float(scale), M, K,
                                           N, _ptr(Y), _ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "mmssl_linear_split_f32")
    return Y


def _split_pair_dev(x):
    """(hi, lo) bf16 pair of a contiguous fp32 tensor in one launch."""
    hi = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    lo = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    rc = _lib.lib().mmssl_split_bf16_f32(_ptr(x), x.numel(), _ptr(hi), _ptr(lo), _lib.stream_ptr())
    _lib.check(rc, "mmssl_split_bf16_f32")
    return hi, lo


def _linear_raw(F_, W, b, keep, scale):
    M, K = F_.shape
    N = W.shape[0]
    if _use_split(F_):
        (Fh, Fl) = _split_features(F_)[0]
        Wh, Wl = _split_pair_dev(W)
        return _split_call(Fh, Fl, Wh, Wl, b, keep, scale, M, K, N)
    Y = torch.empty((M, N), dtype=torch.float32, device=F_.device)
    ft = _FT.get((F_.data_ptr(), tuple(F_.shape))) if fwd_ft_enabled() else None
    if ft is not None:
        nbf = _lib.lib().mmssl_linear_ft_workspace_bytes(M, K, N, ft[1])
        if nbf > 0:
            wsf = torch.empty(nbf // 4, dtype=torch.float32, device=F_.device)
            rc = _lib.lib().mmssl_linear_ft_f32(_ptr(ft[0]), ft[1], _ptr(W), _ptr(b), _ptr(keep), float(scale), M, K, N,
                                                _ptr(Y), _ptr(wsf), nbf, _lib.stream_ptr())
            _lib.check(rc, "mmssl_linear_ft_f32")
            return Y
    nb = _lib.lib().mmssl_linear_workspace_bytes(M, K, N)
    ws = torch.empty(max(nb // 4, 4), dtype=torch.float32, device=F_.device)
    tk = _linear_tickets(F_, M, K, N)
    if tk is not None:
        tk, ws = tk               # the fix-up form owns its partial slots (see _linear_tickets)
        rc = _lib.lib().mmssl_linear_tk_f32(_ptr(F_), _ptr(W), _ptr(b), _ptr(keep), float(scale), M, K, N, _ptr(Y),
                                            _ptr(ws), ws.numel() * 4, _ptr(tk), _lib.stream_ptr())
        _lib.check(rc, "mmssl_linear_tk_f32")
        return Y
    rc = _lib.lib().mmssl_linear_f32(_ptr(F_), _ptr(W), _ptr(b), _ptr(keep), float(scale), M, K, N, _ptr(Y),
                                     _ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "mmssl_linear_f32")
    return Y


# Arrival tickets of the in-kernel stream-K fix-up (mmssl_linear_tk_f32): zero-initialised once, left zero by every
# call, one set per (shape, stream) so that products that run side by side never share one.
_TICKETS = {}


def linear_fixup_enabled():
    """OPT-IN (MMSSL_GEMM_FIXUP=1): the stream-K fix-up inside the kernel instead of the separate reduce launch.
    Measured: Baby image forward alone 102.3 vs 104.7 us, but the whole step 0.641 vs 0.625 ms, so the default stays
    the two-launch form. (The same fix-up for the weight gradient measured 0.604 vs 0.592 ms and was dropped.)"""
    return _os.environ.get("MMSSL_GEMM_FIXUP", "0") == "1"


def _linear_tickets(F_, M, K, N):
    if not linear_fixup_enabled():
        return None
    # launches on one stream are ordered, so one set per (shape, stream) is never in use twice at a time; the sets are
    # kept for the life of the process (a captured graph holds their addresses)
    key = (F_.device.index, M, K, N, torch.cuda.current_stream(F_.device).cuda_stream)
    tk = _TICKETS.get(key)
    if tk is None:
        n = int(_lib.lib().mmssl_linear_ticket_count(M, K, N))
        if n <= 0:
            _TICKETS[key] = False
            return None
        # tickets AND partial slots are private to this key for the life of the process: the fix-up's write-through
        # stores / cache-bypassing loads are not ordered against ordinary cached accesses other kernels may have made to
        # recycled allocator memory (a transient workspace showed order-dependent wrong results in a long test run)
        nb = _lib.lib().mmssl_linear_workspace_bytes(M, K, N)
        tk = (torch.zeros(n, dtype=torch.int32, device=F_.device),
              torch.zeros(max(nb // 4, 4), dtype=torch.float32, device=F_.device))
        _TICKETS[key] = tk
    return tk if tk is not False else None


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


# Step-owned counter ticks (hotpath.HotPathStep): with EXTERNAL["on"] the dropout launch and the fused AdamW do not
# launch their own one-thread counter kernels; the step's loss-assembly launch advances all counters instead
# (loss_assemble(..., ticks=...)). Off by default: every other caller keeps self-advancing launches.
EXTERNAL = {"on": False}


def external_ticks(flag):
    prev = EXTERNAL["on"]
    EXTERNAL["on"] = bool(flag)
    return prev


def dropout_masks(count, rows, cols, p, device):
    """uint8 [count, rows, cols], 1 = keep with probability 1-p; the generator state lives on the device and
    advances by itself, so a captured step draws fresh masks on every replay."""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.MmsslError("dropout_masks: needs a CUDA (HIP) device")
    n = count * rows * cols
    pad = (-n) % 4
    buf = torch.empty(n + pad, dtype=torch.uint8, device=device)
    rc = _lib.lib().mmssl_dropout_mask_ex_u8(_ptr(_rng_state(device)), float(p), n + pad, _ptr(buf),
                                             1 if EXTERNAL["on"] else 0, _lib.stream_ptr())
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
            if gY is None:                 # the transposed-feature path does not return the masked gradient
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


def _linear_wgrad_split(gY0, keep, scale, F_, W):
    M, K = F_.shape
    N = W.shape[0]
    # gW [N, K] = gY^T [N, Mp] . F^T [K, Mp]^T through the same split kernel (reduction over the padded M);
    # one preparation launch transposes, masks and splits gY and sums its columns (bias gradient)
    _, (FTh, FTl), Mp, _keep_alive = _split_features(F_)
    dev = gY0.device
    gTh = torch.empty((N, Mp), dtype=torch.bfloat16, device=dev)
    gTl = torch.empty((N, Mp), dtype=torch.bfloat16, device=dev)
    gb = torch.empty(N, dtype=torch.float32, device=dev)
    nbt = _lib.lib().mmssl_split_transpose_workspace_bytes(Mp, N)
    wst = torch.empty(max(nbt // 4, 4), dtype=torch.float32, device=dev)
    rc = _lib.lib().mmssl_split_transpose_bf16_f32(_ptr(gY0), _ptr(keep), float(scale), M, N, Mp, _ptr(gTh),
                                                   _ptr(gTl), _ptr(gb), _ptr(wst), wst.numel() * 4,
                                                   _lib.stream_ptr())
    _lib.check(rc, "mmssl_split_transpose_bf16_f32")
    gW = _split_call(gTh, gTl, FTh, FTl, None, None, 1.0, N, Mp, K)
    return None, gW, gb


# Weight gradient through the FORWARD kernel: gW [N, K] = gYm^T [N, Mp] . (F^T [K, Mp])^T. The feature matrices are
# constants (Models.py:46-47), so F^T is built once (register_transposed_features, + M*K*4 bytes of HBM); per call one
# kernel transposes + dropout-masks gY and sums its columns (the bias gradient). Both projection GEMMs then run on
# the same stream-K LDS-DMA kernel (csrc/linear.hip, gemm_sk_kernel). Opt-in (see wgrad_ft_enabled); the default is
# the register-staged wgrad kernel (mmssl_linear_wgrad_f32), which is also what unregistered inputs use.
_FT = {}

# Weight gradients as row-range partials (hotpath.HotPathStep): with _WPARTS["on"] the fused weight-gradient path
# returns no gradient tensors; it leaves (workspace, n_parts, weight stride, bias offset, N) under the weight's
# data_ptr for the optimiser, which adds the slices while it reads them.
_WPARTS = {"on": False, "map": {}}


def wgrad_parts(flag):
    prev = _WPARTS["on"]
    _WPARTS["on"] = bool(flag)
    return prev


def take_wgrad_parts():
    m, _WPARTS["map"] = _WPARTS["map"], {}
    return m


def wgrad_ft_enabled():
    """OPT-IN (MMSSL_WGRAD_FT=1). Measured on MI355X (tools/gemm_v6_probe.py, hipGraph replay): Baby image wgrad
    130-135 us vs 138-141 us for the register-staged kernel, Baby text 59 vs 52 us, and no difference in the whole
    step (0.623 vs 0.620 ms) for +376 MB of HBM, so the default stays the register-staged kernel."""
    return _os.environ.get("MMSSL_WGRAD_FT", "0") == "1"


def fwd_ft_enabled():
    """OPT-IN (MMSSL_FWD_FT=1): the projection forward from the transposed feature copy (mmssl_linear_ft_f32) for
    matrices registered with register_transposed_features."""
    return _os.environ.get("MMSSL_FWD_FT", "0") == "1"


def register_transposed_features(F_):
    key = (F_.data_ptr(), tuple(F_.shape))
    hit = _FT.get(key)
    if hit is None:
        M, K = F_.shape
        Mp = (M + 63) // 64 * 64
        FT = torch.zeros((K, Mp), dtype=torch.float32, device=F_.device)
        FT[:, :M] = F_.t()
        hit = (FT, Mp, F_)                      # keeps F_ alive: the key stays valid
        _FT[key] = hit
    return hit


def _linear_wgrad_ft(gY, keep, scale, F_, W, FT, Mp):
    M, K = F_.shape
    N = W.shape[0]
    dev = gY.device
    gT = torch.empty((N, Mp), dtype=torch.float32, device=dev)
    gb = torch.empty(N, dtype=torch.float32, device=dev)
    nbt = _lib.lib().mmssl_transpose_mask_workspace_bytes(Mp, N)
    wst = torch.empty(max(nbt // 4, 4), dtype=torch.float32, device=dev)
    rc = _lib.lib().mmssl_transpose_mask_f32(_ptr(gY), _ptr(keep), float(scale), M, N, Mp, _ptr(gT), _ptr(gb),
                                             _ptr(wst), wst.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "mmssl_transpose_mask_f32")
    gW = _linear_raw(gT, FT, None, None, 1.0)          # [N, Mp] x [K, Mp]^T -> [N, K]
    return None, gW, gb


def _linear_wgrad_raw(gY, keep, scale, F_, W):
    """(masked gY or None, gW, gb) for Y = dropout(F W^T + b): dropout backward, then the wgrad GEMM."""
    M, K = F_.shape
    N = W.shape[0]
    gY0 = gY
    if _use_split(F_):
        return _linear_wgrad_split(gY0, keep, scale, F_, W)
    ft = _FT.get((F_.data_ptr(), tuple(F_.shape))) if wgrad_ft_enabled() else None
    if ft is not None and N % 4 == 0 and N <= 256:
        return _linear_wgrad_ft(gY0, keep, scale, F_, W, ft[0], ft[1])
    fused = keep is not None and _lib.lib().mmssl_linear_wgrad_fuses_mask(M, K, N) == 1
    if fused and _WPARTS["on"]:
        # the caller's optimiser adds the row-range partials itself (mmssl_adamw_sliced_f32): no reduce launch, no
        # materialised gW / gb; the partial buffers are handed over through take_wgrad_parts()
        nb = _lib.lib().mmssl_linear_wgrad_workspace_bytes(M, K, N)
        ws = torch.empty(max(nb // 4, 4), dtype=torch.float32, device=W.device)
        n_parts, b_off = _ct.c_int(0), _ct.c_int64(0)
        rc = _lib.lib().mmssl_linear_wgrad_parts_f32(_ptr(gY), _ptr(keep), float(scale), _ptr(F_), M, K, N, _ptr(ws), nb,
                                                     _ct.byref(n_parts), _ct.byref(b_off), _lib.stream_ptr())
        _lib.check(rc, "mmssl_linear_wgrad_parts_f32")
        _WPARTS["map"][W.data_ptr()] = (ws, int(n_parts.value), N * K, int(b_off.value), N)
        return None, None, None
    if fused:                   # register-direct kernel: dropout backward + bias gradient on the loaded fragments
        gW = torch.empty_like(W)
        gb = torch.empty(N, dtype=torch.float32, device=W.device)
        nb = _lib.lib().mmssl_linear_wgrad_workspace_bytes(M, K, N)
        ws = torch.empty(max(nb // 4, 4), dtype=torch.float32, device=W.device)
        rc = _lib.lib().mmssl_linear_wgrad_f32(_ptr(gY), _ptr(keep), float(scale), _ptr(F_), M, K, N, _ptr(gW), _ptr(gb),
                                               _ptr(ws), nb, _lib.stream_ptr())
        _lib.check(rc, "mmssl_linear_wgrad_f32")
        return None, gW, gb
    if keep is not None:        # register-staged kernel: one dropout-backward pass first (its in-fetch variant is slower)
        gYm = torch.empty_like(gY)
        rc = _lib.lib().mmssl_mask_scale_f32(_ptr(gY), _ptr(keep), float(scale), gY.numel(), _ptr(gYm),
                                             _lib.stream_ptr())
        _lib.check(rc, "mmssl_mask_scale_f32")
        gY = gYm
    gW = torch.empty_like(W)
    gb = torch.empty(N, dtype=torch.float32, device=W.device)
    nb = _lib.lib().mmssl_linear_wgrad_workspace_bytes(M, K, N)
    ws = torch.empty(max(nb // 4, 4), dtype=torch.float32, device=W.device)
    rc = _lib.lib().mmssl_linear_wgrad_f32(_ptr(gY), None, 1.0, _ptr(F_), M, K, N, _ptr(gW), _ptr(gb), _ptr(ws), nb,
                                           _lib.stream_ptr())
    _lib.check(rc, "mmssl_linear_wgrad_f32")
    return gY, gW, gb


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
    out = (torch.zeros if ld != n else torch.empty)((B, ld), dtype=torch.float32, device=dev)
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


def topk_rows(X, k, values=False):
    """Columns of the k largest entries of every row, descending value, ties by ascending column (the order
    heapq.nlargest gives the reference's evaluation, batch_test.py:21-36). X may be a row-pitched view."""
    if not (X.is_cuda and X.dtype == torch.float32 and X.dim() == 2 and X.stride(1) == 1):
        raise _lib.MmsslError("topk_rows: fp32 [B, n] HIP tensor with unit column stride expected")
    B, n = X.shape
    idx = torch.empty((B, k), dtype=torch.int64, device=X.device)
    val = torch.empty((B, k), dtype=torch.float32, device=X.device) if values else None
    rc = _lib.lib().mmssl_topk_rows_f32(_ptr(X), B, n, X.stride(0), int(k), _ptr(idx), _ptr(val), _lib.stream_ptr())
    _lib.check(rc, "mmssl_topk_rows_f32")
    return (idx, val) if values else idx


def rows_membership(rowptr, cols, rows, cand):
    """uint8 [B, K]: cand[b, k] in CSR row rows[b] (int32 rowptr / sorted cols on the device)."""
    B, K = cand.shape
    out = torch.empty((B, K), dtype=torch.uint8, device=cand.device)
    rc = _lib.lib().mmssl_rows_membership_u8(_ptr(rowptr), _ptr(cols), _ptr(rows), B, K, _ptr(cand.contiguous()), _ptr(out),
                                             _lib.stream_ptr())
    _lib.check(rc, "mmssl_rows_membership_u8")
    return out


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
    def forward(ctx, ua, ia, img_uid, txt_uid, users, pos, neg, decay, batch_size, tau, overlap=None, eager_w=None,
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
                                               eager_w, tail, out, wsb, nb)
        ctx.eager = None
        # the table gradients the backward scatter-adds into are allocated here and zero-filled on a side
        # stream, next to the BPR forward, while the (longer) InfoNCE forward runs on the current stream
        g_ua = torch.empty_like(ua) if need_grad else None
        g_ia = torch.empty_like(ia) if need_grad else None
        overlap = loss_overlap_enabled() if overlap is None else bool(overlap)
        main = torch.cuda.current_stream(dev)
        side = _side_streams(dev)[0] if overlap else main
        if overlap:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            rc = _lib.lib().mmssl_bpr_fwd_f32(_ptr(ua), _ptr(ia), None, _ptr(users), _ptr(pos), _ptr(neg), B, d,
                                              float(decay), int(batch_size), _ptr(out), _ptr(wsb), nb,
                                              _lib.stream_ptr())
            _lib.check(rc, "mmssl_bpr_fwd_f32")
            if need_grad:
                g_ua.zero_()
                g_ia.zero_()
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
        if overlap:
            main.wait_stream(side)
        ctx.save_for_backward(ua, ia, users, pos, neg, ws1)
        ctx.gbuf = (g_ua, g_ia)
        ctx.cfg = (B, d, float(decay), int(batch_size), float(tau), img_uid.shape, txt_uid.shape, overlap)
        return out

    @staticmethod
    def _forward_eager(ctx, ua, ia, img_uid, txt_uid, users, pos, neg, decay, batch_size, tau, w, tail, out, wsb, nb):
        """The caller PROMISES that the result is backpropagated with exactly the gradient `w` (a persistent [5]
        tensor: HotPathStep's loss weights, total = w . terms + c * extra backpropagated with 1). The gradients of
        the loss terms are then known before the loss scalars are, and the whole loss section becomes ONE chain of
        seven launches on the current stream, no fork / join (a cross-queue edge of a replayed hipGraph costs 10-15 us):
          zero fill (all four gradients + the tickets), InfoNCE prep, pair tiles, row terms (the last block also reduces
          the two losses), backward pair tiles, backward finish, BPR backward + BPR loss + loss assembly + counter ticks.
        tail = (extra, c, total, ticks): see mmssl_bpr_step_f32. backward() returns the stored gradients."""
        B, d = users.shape[0], ua.shape[1]
        dev = ua.device
        if w.dtype != torch.float32 or w.numel() != 5 or w.device != dev:
            raise _lib.MmsslError("batch_losses: eager_w must be a [5] fp32 tensor on the tables' device")
        extra, c, total, ticks = tail
        n_ua, n_ia, n_im, n_tx = ua.numel(), ia.numel(), img_uid.numel(), txt_uid.numel()
        gbuf, _PREFILL["buf"] = _PREFILL["buf"], None          # zero-filled by the forward (see _PREFILL), if enabled
        if gbuf is None or gbuf.numel() != n_ua + n_ia + n_im + n_tx + 4 or gbuf.device != dev:
            gbuf = torch.zeros(n_ua + n_ia + n_im + n_tx + 4, dtype=torch.float32, device=dev)
        g_ua = gbuf[:n_ua].view_as(ua)
        g_ia = gbuf[n_ua:n_ua + n_ia].view_as(ia)
        g_img = gbuf[n_ua + n_ia:n_ua + n_ia + n_im].view_as(img_uid)
        g_txt = gbuf[n_ua + n_ia + n_im:n_ua + n_ia + n_im + n_tx].view_as(txt_uid)
        tickets = gbuf[n_ua + n_ia + n_im + n_tx:]              # three zeroed ints (InfoNCE x2, BPR)
        nbw = _lib.lib().mmssl_infonce_multi_workspace_bytes(2, B, d)
        if nbw == 0:
            raise _lib.MmsslError("infonce: unsupported shape n=%d d=%d" % (B, d))
        ws1 = torch.empty(nbw // 4, dtype=torch.float32, device=dev)
        z1s = (_ct.c_void_p * 2)(img_uid.data_ptr(), txt_uid.data_ptr())
        rc = _lib.lib().mmssl_infonce_multi_fwd_ticket_f32(z1s, _ptr(ua), _ptr(users), 2, B, d, float(tau), _ptr(out[3:5]),
                                                           _ptr(ws1), nbw, _ptr(tickets), _lib.stream_ptr())
        _lib.check(rc, "mmssl_infonce_multi_fwd_ticket_f32")
        gz1s = (_ct.c_void_p * 2)(_ptr(g_img), _ptr(g_txt))
        f32s, u64s = ticks if ticks else ((), ())
        fa = (_ct.c_void_p * max(len(f32s), 1))(*[int(x) for x in f32s])
        ka = (_ct.c_void_p * max(len(u64s), 1))(*[int(x) for x in u64s])
        xparts, n_xparts = None, 0
        last = _DEFER_SS["last"]
        if last is not None and extra is not None and last[0].data_ptr() == extra.data_ptr():
            xparts, n_xparts = last[1], last[1].numel()       # `extra` is a forward's unreduced regulariser sum
            _DEFER_SS["last"] = None
        if d <= 64 and _os.environ.get("MMSSL_BPR_GUEST", "1") == "1":
            # the BPR tail as guest blocks of the InfoNCE backward pair-tile launch (it depends on nothing in it), then
            # the InfoNCE finish: six launches, and the chain is shorter by the BPR tail's whole duration
            rc = _lib.lib().mmssl_infonce_bwd_tiles_bpr_f32(
                _ptr(users), 2, B, d, float(tau), _ptr(w[3:5]), gz1s, _ptr(g_ua), _ptr(ws1), ws1.numel() * 4,
                _ptr(ua), _ptr(ia), _ptr(users), _ptr(pos), _ptr(neg), B, float(decay), int(batch_size), _ptr(w[0:1]),
                _ptr(w[1:2]), _ptr(g_ua), _ptr(g_ia), _ptr(out), _ptr(w), 5, _ptr(extra), float(c), _ptr(total), fa,
                len(f32s), ka, len(u64s), _ptr(wsb), nb, _ptr(tickets[2:]), _ptr(xparts), n_xparts, _lib.stream_ptr())
            _lib.check(rc, "mmssl_infonce_bwd_tiles_bpr_f32")
            rc = _lib.lib().mmssl_infonce_multi_bwd_phase_f32(_ptr(users), 2, B, d, float(tau), _ptr(w[3:5]), gz1s,
                                                              _ptr(g_ua), _ptr(ws1), ws1.numel() * 4, 2, _lib.stream_ptr())
            _lib.check(rc, "mmssl_infonce_multi_bwd_phase_f32")
            ctx.eager = (g_ua, g_ia, g_img if ctx.needs_input_grad[2] else None, g_txt if ctx.needs_input_grad[3] else None)
            return out
        rc = _lib.lib().mmssl_infonce_multi_bwd_phase_f32(_ptr(users), 2, B, d, float(tau), _ptr(w[3:5]), gz1s, _ptr(g_ua),
                                                          _ptr(ws1), ws1.numel() * 4, 3, _lib.stream_ptr())
        _lib.check(rc, "mmssl_infonce_multi_bwd_phase_f32")
        rc = _lib.lib().mmssl_bpr_step_f32(_ptr(ua), _ptr(ia), _ptr(users), _ptr(pos), _ptr(neg), B, d, float(decay),
                                           int(batch_size), _ptr(w[0:1]), _ptr(w[1:2]), _ptr(g_ua), _ptr(g_ia), _ptr(out),
                                           _ptr(w), 5, _ptr(extra), float(c), _ptr(total), fa, len(f32s), ka, len(u64s),
                                           _ptr(wsb), nb, _ptr(tickets[2:]), _ptr(xparts), n_xparts, _lib.stream_ptr())
        _lib.check(rc, "mmssl_bpr_step_f32")
        ctx.eager = (g_ua, g_ia, g_img if ctx.needs_input_grad[2] else None, g_txt if ctx.needs_input_grad[3] else None)
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.eager is not None:
            g_ua, g_ia, g_img, g_txt = ctx.eager
            ctx.eager = None
            return g_ua, g_ia, g_img, g_txt, None, None, None, None, None, None, None, None, None
        ua, ia, users, pos, neg, ws1 = ctx.saved_tensors
        B, d, decay, batch_size, tau, s_img, s_txt, overlap = ctx.cfg
        g = g.contiguous().to(torch.float32)
        g_ua, g_ia = ctx.gbuf
        ctx.gbuf = None
        if g_ua is None:                     # second backward through a retained graph
            g_ua, g_ia = torch.zeros_like(ua), torch.zeros_like(ia)
        dev = ua.device
        g_img = torch.zeros(s_img, dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None
        g_txt = torch.zeros(s_txt, dtype=torch.float32, device=dev) if ctx.needs_input_grad[3] else None
        gz1s = (_ct.c_void_p * 2)(_ptr(g_img), _ptr(g_txt))
        main = torch.cuda.current_stream(dev)
        side = _side_streams(dev)[0] if overlap else main
        if overlap:
            side.wait_stream(main)
        # BPR backward (scatter-add into g_ua / g_ia) next to the InfoNCE pair tiles (workspace only);
        # the InfoNCE finish scatter-adds into g_ua AFTER the join: same accumulation order as serial
        with torch.cuda.stream(side):
            rc = _lib.lib().mmssl_bpr_bwd_f32(_ptr(ua), _ptr(ia), None, _ptr(users), _ptr(pos), _ptr(neg), B, d,
                                              decay, batch_size, _ptr(g[0:1]), _ptr(g[1:2]), _ptr(g_ua), _ptr(g_ia),
                                              None, _lib.stream_ptr())
            _lib.check(rc, "mmssl_bpr_bwd_f32")
        call = _lib.lib().mmssl_infonce_multi_bwd_phase_f32
        rc = call(_ptr(users), 2, B, d, tau, _ptr(g[3:5]), gz1s, _ptr(g_ua), _ptr(ws1), ws1.numel() * 4, 1,
                  _lib.stream_ptr())
        _lib.check(rc, "mmssl_infonce_multi_bwd_phase_f32")
        if overlap:
            main.wait_stream(side)
        rc = call(_ptr(users), 2, B, d, tau, _ptr(g[3:5]), gz1s, _ptr(g_ua), _ptr(ws1), ws1.numel() * 4, 2,
                  _lib.stream_ptr())
        _lib.check(rc, "mmssl_infonce_multi_bwd_phase_f32")
        return g_ua, g_ia, g_img, g_txt, None, None, None, None, None, None, None, None, None


def eager_loss_backward_enabled():
    """MMSSL_EAGER_LOSS_BWD=0: the loss backward waits for autograd (the round-1 structure)."""
    return _os.environ.get("MMSSL_EAGER_LOSS_BWD", "1") == "1"


def batch_losses_vec(ua, ia, img_uid, txt_uid, users, pos, neg, decay, batch_size, tau, overlap=None, eager_w=None,
                     tail=None):
    """[mf_loss, emb_loss, 0, cl_img, cl_txt] as ONE tensor (see _BatchLosses / loss_assemble).
    overlap=False keeps every launch on the current stream (None: the MMSSL_STREAMS default).
    eager_w + tail=(extra, c, total, ticks): the caller's promise that the result is backpropagated with exactly the
    [5] gradient eager_w, total = eager_w . terms + c * extra being written to `total` by the same launches
    (see _BatchLosses._forward_eager)."""
    dev = ua.device
    if eager_w is not None and (tail is None or not eager_loss_backward_enabled()):
        eager_w = tail = None
    return _BatchLosses.apply(ua, ia, img_uid, txt_uid, _idx(users, "users", dev), _idx(pos, "pos", dev),
                              _idx(neg, "neg", dev), decay, batch_size, tau, overlap, eager_w, tail)


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
    still sees a zero — not a missing — gradient for `w`, like the reference's autograd)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.wshape, ctx.wdev = w.shape, w.device
        ctx.lazy = _ANCHOR["lazy"]
        if ctx.lazy:
            _ANCHOR["params"].append(w)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if ctx.lazy:          # the caller assigns the (persistent) zero gradient itself: no fill launch in the backward
            return g, None
        return g, torch.zeros(ctx.wshape, dtype=torch.float32, device=ctx.wdev)


# Lazy anchors (hotpath.HotPathStep): the anchor's backward launches nothing; after the backward the step points the
# `.grad` of every anchored parameter that got no gradient at a persistent all-zero tensor (take_anchored_params).
_ANCHOR = {"lazy": False, "params": [], "zeros": {}}


def lazy_anchors(flag):
    prev = _ANCHOR["lazy"]
    _ANCHOR["lazy"] = bool(flag)
    return prev


def assign_anchored_zero_grads():
    """Give every parameter anchored since the last call an exactly-zero gradient if autograd produced none."""
    params, _ANCHOR["params"] = _ANCHOR["params"], []
    for w in params:
        if w.grad is None:
            key = (w.data_ptr(), tuple(w.shape))
            z = _ANCHOR["zeros"].get(key)
            if z is None:
                z = torch.zeros_like(w)
                _ANCHOR["zeros"][key] = z
            w.grad = z


def zero_grad_anchor(x, w):
    return _ZeroGradAnchor.apply(x, w)


# ---------------------------------------------------------------------------------------
# GCN propagation + layer mean + modality fusion as ONE autograd node       Models.py:199-218
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


class _PropagateFuse(torch.autograd.Function):
    """Everything after the projection, as one autograd node (Models.py:177-178,182-183,199-218):
         img_user = A_ui.x_img, img_item = A_iu.img_user          (and the text pair)
         u_l = A_ui.i_{l-1}, i_l = A_iu.u_l (row softmax on the last layer)
         u_g = mean_l(u_l) + r*normalize(img_user) + r*normalize(txt_user)        (items alike)
         ss  = |img_user|^2+|txt_user|^2+|img_item|^2+|txt_item|^2 (feature-regulariser sum)
    Forward: 4 + 2G SpMM + 2 combine + 1 tiny reduce. Backward: 2 combine + 1 softmax-bwd + 4 + 2G SpMM
    whose epilogues add the other gradient branch / the layer-mean gradient and apply the softmax
    backward, so no separate accumulation or scaling kernels run."""

    @staticmethod
    def forward(ctx, u0, i0, x_img, x_txt, ui, iu, n_layers, r):
        u0, i0, x_img, x_txt = _chk(u0, "u0"), _chk(i0, "i0"), _chk(x_img, "x_img"), _chk(x_txt, "x_txt")
        img_user = _spmm_raw(ui, False, x_img, EPI_NONE)
        img_item = _spmm_raw(iu, False, img_user, EPI_NONE)
        txt_user = _spmm_raw(ui, False, x_txt, EPI_NONE)
        txt_item = _spmm_raw(iu, False, txt_user, EPI_NONE)
        us, its = [u0], [i0]
        u, i = u0, i0
        for l in range(n_layers):
            epi = EPI_SOFTMAX if l == n_layers - 1 else EPI_NONE
            u = _spmm_raw(ui, False, i, epi)
            i = _spmm_raw(iu, False, u, epi)
            us.append(u)
            its.append(i)
        inv = 1.0 / (n_layers + 1)
        d = u0.shape[1]
        nbu = _lib.lib().mmssl_layer_combine_blocks(u0.shape[0], d)
        nbi = _lib.lib().mmssl_layer_combine_blocks(i0.shape[0], d)
        part = torch.empty(nbu + nbi, dtype=torch.float32, device=u0.device)
        u_g = _combine_fwd(us, inv, img_user, txt_user, r, part[:nbu])
        i_g = _combine_fwd(its, inv, img_item, txt_item, r, part[nbu:])
        ss = torch.empty((), dtype=torch.float32, device=u0.device)
        rc = _lib.lib().mmssl_sum_partials_f32(_ptr(part), nbu + nbi, _ptr(ss), _lib.stream_ptr())
        _lib.check(rc, "mmssl_sum_partials_f32")
        ctx.save_for_backward(img_user, txt_user, img_item, txt_item, us[-1], its[-1])
        ctx.cfg = (ui, iu, n_layers, float(r), inv)
        ctx.set_materialize_grads(False)     # unused outputs arrive as None, not as zero-filled tensors
        return u_g, i_g, ss, img_item, txt_item, img_user, txt_user

    @staticmethod
    def backward(ctx, Gu, Gi, g_ss, G_img_item, G_txt_item, G_img_user, G_txt_user):
        img_user, txt_user, img_item, txt_item, uG, iG = ctx.saved_tensors
        ui, iu, n_layers, r, inv = ctx.cfg
        Gu = _chk(Gu, "Gu") if Gu is not None else torch.zeros_like(img_user)
        Gi = _chk(Gi, "Gi") if Gi is not None else torch.zeros_like(img_item)
        g_ss = g_ss.contiguous().to(torch.float32) if g_ss is not None else None
        # d(ss)/dx = 2x; normalize-backward and the regulariser term share one pass over A, B
        g_iu_, g_tu_, g_u0 = _combine_bwd(img_user, txt_user, Gu, r, inv, g_ss, 2.0, True)
        g_ii_, g_ti_, _ = _combine_bwd(img_item, txt_item, Gi, r, inv, g_ss, 2.0, False)
        # gradients that arrive on the modal outputs themselves (only if a caller used them elsewhere)
        if G_img_item is not None:
            g_ii_ = g_ii_ + G_img_item
        if G_txt_item is not None:
            g_ti_ = g_ti_ + G_txt_item
        if G_img_user is not None:
            g_iu_ = g_iu_ + G_img_user
        if G_txt_user is not None:
            g_tu_ = g_tu_ + G_txt_user
        # modal chains: g(img_user) = A_iu^T g(img_item) + own branch (AXPY epilogue); g(x) = A_ui^T g(img_user)
        g_x_img = _spmm_raw(ui, True, _spmm_raw(iu, True, g_ii_, EPI_AXPY, g_iu_, 1.0), EPI_NONE)
        g_x_txt = _spmm_raw(ui, True, _spmm_raw(iu, True, g_ti_, EPI_AXPY, g_tu_, 1.0), EPI_NONE)
        # GCN chain. last layer: i_G only feeds the mean; u_G feeds the mean and A_iu.u_G
        gi = softmax_rows_bwd(iG, Gi, inv)
        gu = _spmm_raw(iu, True, gi, EPI_AXPY_SOFTMAX_BWD, Gu, inv, uG)
        gi = _spmm_raw(ui, True, gu, EPI_AXPY, Gi, inv)            # total gradient of i_{G-1}
        for _ in range(n_layers - 1):
            gu = _spmm_raw(iu, True, gi, EPI_AXPY, Gu, inv)
            gi = _spmm_raw(ui, True, gu, EPI_AXPY, Gi, inv)
        return g_u0, gi, g_x_img, g_x_txt, None, None, None, None


def propagate_fuse(ui, iu, u0, i0, x_img, x_txt, n_layers, r):
    """(u_g, i_g, ss, img_item, txt_item, img_user, txt_user): see _PropagateFuse. `ui`, `iu` are
    GraphPlans; x_img / x_txt are the projected (and dropped-out) modality features [n_items, d]."""
    out = _PropagateFuse.apply(u0, i0, x_img, x_txt, ui, iu, int(n_layers), float(r))
    return out


# ---------------------------------------------------------------------------------------
# projection + modal chains + GCN chain + fusion as ONE node, three forked streams
# ---------------------------------------------------------------------------------------
import os as _os

_SIDE_STREAMS = {}


def _side_streams(device, n=3):
    key = (device.type, device.index)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        prio = [int(x) for x in _os.environ.get("MMSSL_STREAM_PRIO", "0,0,0").split(",")]
        st = [torch.cuda.Stream(device=device, priority=prio[k % len(prio)]) for k in range(n + 1)]
        _SIDE_STREAMS[key] = st
    return st[:n]


def scalar_stream(device):
    """A fourth forked stream for launches nothing on the step's critical path waits for (loss scalars)."""
    device = torch.device(device)
    _side_streams(device)
    return _SIDE_STREAMS[(device.type, device.index)][3]


def _branch_order(var, default):
    """Enqueue order of the three chains (A image, B text, C GCN): the order in which a captured graph's
    branches were recorded influences how the hipGraph executor interleaves them."""
    o = _os.environ.get(var, default).upper()
    return o if sorted(o) == ["A", "B", "C"] else default


def overlap_enabled():
    return _os.environ.get("MMSSL_STREAMS", "1") != "0"


def loss_overlap_enabled():
    """BPR next to InfoNCE on a forked stream inside the loss node (MMSSL_LOSS_OVERLAP=0: one stream). Measured on the
    Baby step under hipGraph replay: forked 0.636 ms, one stream 0.654 ms."""
    return overlap_enabled() and _os.environ.get("MMSSL_LOSS_OVERLAP", "1") == "1"


# Deferred join of the weight-gradient chains. By default _HotForward.backward returns with every side
# stream joined. A caller that owns the whole step (hotpath.HotPathStep) may set this flag: backward then
# returns as soon as the embedding-table gradients are complete, the projection wgrad GEMMs still running on
# their side streams, and the caller must call join_side_streams() before anything reads image_trans /
# text_trans gradients (it updates the embedding tables in between).
_DEFER = {"on": False}

# Deferred regulariser sum. A caller whose loss tail is mmssl_bpr_step_f32 (hotpath.HotPathStep) may set this flag:
# _HotForward.forward then returns `ss` UNREDUCED (its partial sums are handed to the tail through _DEFER_SS["last"],
# which reduces them and stores the value into the same tensor), and one launch leaves the front of the loss chain.
_DEFER_SS = {"on": False, "last": None}


# Pre-filled loss buffer: with _PREFILL["floats"] set (a function (n_users, n_items, d) -> float count), the forward
# zero-fills a buffer of that size on the GCN chain's stream and leaves it in _PREFILL["buf"] for the loss tail.
_PREFILL = {"floats": None, "buf": None}


def prefill_loss_buffer(fn):
    prev = _PREFILL["floats"]
    _PREFILL["floats"] = fn
    if fn is None:
        _PREFILL["buf"] = None
    return prev


def defer_feat_sumsq(flag):
    prev = _DEFER_SS["on"]
    _DEFER_SS["on"] = bool(flag)
    if not flag:
        _DEFER_SS["last"] = None
    return prev


def defer_wgrad_join(flag):
    prev = _DEFER["on"]
    _DEFER["on"] = bool(flag)
    return prev


def join_side_streams(device):
    """Make the current stream wait for everything queued on this package's side streams."""
    device = torch.device(device)
    main = torch.cuda.current_stream(device)
    for st in _SIDE_STREAMS.get((device.type, device.index), ()):
        main.wait_stream(st)


class _HotForward(torch.autograd.Function):
    """x_m = dropout(F_m W_m^T + b_m) -> modal SpMM chains -> G-layer GCN -> layer mean + modality
    fusion (+ regulariser sum): the complete MMSSL.forward after the id-embedding fusion
    (Models.py:173-174,177-178,182-183,199-218) as one autograd node.

    Three chains are independent and bound by different resources — image projection + its modal chain
    (fp32 MFMA, then gather), text projection + chain, and the GCN chain (gather, latency) — so the
    node forks three HIP streams from the current one, runs one chain on each and joins before the
    combine kernels; the backward mirrors that (GCN backward || image wgrad path || text wgrad path).
    Every side-stream region starts with wait_stream(current) and ends joined into the current
    stream, so the node looks single-stream from outside (autograd, caching allocator, hipGraph
    capture all see ordinary fork/join edges)."""

    @staticmethod
    def forward(ctx, F_img, W_img, b_img, keep_img, F_txt, W_txt, b_txt, keep_txt, scale, u0, i0, ui, iu,
                n_layers, r, overlap):
        F_img, W_img, F_txt, W_txt = _chk(F_img, "F_img"), _chk(W_img, "W_img"), _chk(F_txt, "F_txt"), _chk(W_txt, "W_txt")
        u0, i0 = _chk(u0, "u0"), _chk(i0, "i0")
        dev = u0.device
        main = torch.cuda.current_stream(dev)
        sA, sB, sC = _side_streams(dev) if overlap else (main, main, main)
        if overlap:
            for st in (sA, sB, sC):
                st.wait_stream(main)
        out = {}

        def chain_a():
            with torch.cuda.stream(sA):
                x_img = _linear_raw(F_img, W_img, b_img, keep_img, scale)
                out["img_user"] = _spmm_raw(ui, False, x_img, EPI_NONE)
                out["img_item"] = _spmm_raw(iu, False, out["img_user"], EPI_NONE)

        def chain_b():
            with torch.cuda.stream(sB):
                x_txt = _linear_raw(F_txt, W_txt, b_txt, keep_txt, scale)
                out["txt_user"] = _spmm_raw(ui.twin(), False, x_txt, EPI_NONE)
                out["txt_item"] = _spmm_raw(iu.twin(), False, out["txt_user"], EPI_NONE)

        def chain_c():
            with torch.cuda.stream(sC):
                us, its = [u0], [i0]
                u, i = u0, i0
                uic, iuc = ui.twin(2), iu.twin(2)
                for l in range(n_layers):
                    epi = EPI_SOFTMAX if l == n_layers - 1 else EPI_NONE
                    u = _spmm_raw(uic, False, i, epi)
                    i = _spmm_raw(iuc, False, u, epi)
                    us.append(u)
                    its.append(i)
                out["us"], out["its"] = us, its
                if overlap and _PREFILL["floats"] is not None:
                    # the loss section's zero-filled gradient buffer: the GCN chain's stream is idle from here to the
                    # join (the modal chains end later), so the fill costs nothing on the critical path
                    buf = torch.zeros(_PREFILL["floats"](u0.shape[0], i0.shape[0], u0.shape[1]), dtype=torch.float32,
                                      device=dev)
                    buf.record_stream(main)
                    _PREFILL["buf"] = buf

        chains = {"A": chain_a, "B": chain_b, "C": chain_c}
        for c in _branch_order("MMSSL_FWD_ORDER", "ABC"):
            chains[c]()
        img_user, img_item, txt_user, txt_item = out["img_user"], out["img_item"], out["txt_user"], out["txt_item"]
        us, its = out["us"], out["its"]
        if overlap:
            for st in (sA, sB, sC):
                main.wait_stream(st)
        inv = 1.0 / (n_layers + 1)
        d = u0.shape[1]
        nbu = _lib.lib().mmssl_layer_combine_blocks(u0.shape[0], d)
        nbi = _lib.lib().mmssl_layer_combine_blocks(i0.shape[0], d)
        part = torch.empty(nbu + nbi, dtype=torch.float32, device=dev)
        # the two combines are independent: item side next to the user side (MMSSL_COMBINE_FORK=0: one stream; measured
        # equal within noise on the Baby step, 0.654 vs 0.656 ms)
        if overlap and _os.environ.get("MMSSL_COMBINE_FORK", "1") == "1":
            sA.wait_stream(main)
            with torch.cuda.stream(sA):
                i_g = _combine_fwd(its, inv, img_item, txt_item, r, part[nbu:])
            u_g = _combine_fwd(us, inv, img_user, txt_user, r, part[:nbu])
            main.wait_stream(sA)
        else:
            u_g = _combine_fwd(us, inv, img_user, txt_user, r, part[:nbu])
            i_g = _combine_fwd(its, inv, img_item, txt_item, r, part[nbu:])
        ss = torch.empty((), dtype=torch.float32, device=dev)
        if _DEFER_SS["on"]:
            # the caller's loss tail (mmssl_bpr_step_f32) reduces the partials and stores the sum into `ss`
            _DEFER_SS["last"] = (ss, part)
        else:
            rc = _lib.lib().mmssl_sum_partials_f32(_ptr(part), nbu + nbi, _ptr(ss), _lib.stream_ptr())
            _lib.check(rc, "mmssl_sum_partials_f32")
        ctx.save_for_backward(F_img, W_img, keep_img, F_txt, W_txt, keep_txt, img_user, txt_user, img_item, txt_item,
                              us[-1], its[-1])
        ctx.cfg = (ui, iu, n_layers, float(r), inv, float(scale), bool(overlap), b_img is not None, b_txt is not None)
        ctx.set_materialize_grads(False)
        return u_g, i_g, ss, img_item, txt_item, img_user, txt_user

    @staticmethod
    def backward(ctx, Gu, Gi, g_ss, G_img_item, G_txt_item, G_img_user, G_txt_user):
        (F_img, W_img, keep_img, F_txt, W_txt, keep_txt, img_user, txt_user, img_item, txt_item, uG,
         iG) = ctx.saved_tensors
        ui, iu, n_layers, r, inv, scale, overlap, has_bi, has_bt = ctx.cfg
        Gu = _chk(Gu, "Gu") if Gu is not None else torch.zeros_like(img_user)
        Gi = _chk(Gi, "Gi") if Gi is not None else torch.zeros_like(img_item)
        g_ss = g_ss.contiguous().to(torch.float32) if g_ss is not None else None
        dev = Gu.device
        main = torch.cuda.current_stream(dev)
        sA, sB, sC = _side_streams(dev) if overlap else (main, main, main)
        extra = (G_img_item, G_txt_item, G_img_user, G_txt_user)
        # normalise-backward + regulariser gradient of the user side on sA and of the item side on sB, next to each
        # other and next to the GCN chain (which only needs Gu / Gi); each modal chain then needs one tensor from the
        # other stream. (Measured, Baby step under hipGraph replay: this form 0.62 ms; both combines on the current
        # stream followed by ONE fork 0.65 ms - the extra cross-stream hops cost less than the serialised kernels.)
        split = overlap and all(t is None for t in extra)
        if overlap:
            for st in (sA, sB, sC):
                st.wait_stream(main)
            for t, st in ((keep_img, sA), (keep_txt, sB), (img_user, sA), (txt_user, sA), (img_item, sB),
                          (txt_item, sB), (uG, sC), (iG, sC), (Gu, sA), (Gu, sC), (Gi, sB), (Gi, sC)):
                if t is not None and _os.environ.get("MMSSL_NO_RECORD_STREAM") != "1":
                    t.record_stream(st)     # main-pool tensors read on side streams, possibly after this backward returned
        out = {}

        def chain_c():
            with torch.cuda.stream(sC):
                uic, iuc = ui.twin(2), iu.twin(2)
                gi = softmax_rows_bwd(iG, Gi, inv)
                gu = _spmm_raw(iuc, True, gi, EPI_AXPY_SOFTMAX_BWD, Gu, inv, uG)
                gi = _spmm_raw(uic, True, gu, EPI_AXPY, Gi, inv)
                for _ in range(n_layers - 1):
                    gu = _spmm_raw(iuc, True, gi, EPI_AXPY, Gu, inv)
                    gi = _spmm_raw(uic, True, gu, EPI_AXPY, Gi, inv)
                out["gi"] = gi

        c_first = overlap and _os.environ.get("MMSSL_BWD_C_FIRST", "0") == "1"
        if c_first:
            # experiment (MMSSL_BWD_C_FIRST=1): record the GCN chain before the combines. It then starts 70 us earlier,
            # but runs next to the weight-gradient GEMMs for longer and both stretch (wgrad 122 -> 174 us, SpMM 16 ->
            # 50-60 us): 0.640 vs 0.633 ms per Baby step, so the default keeps it after the exchange.
            chain_c()
        if split and _os.environ.get("MMSSL_COMBINE2", "0") == "1":
            # OPT-IN experiment: both sides in ONE launch on the current stream, then the forks (one launch and two
            # cross-queue hops less than a launch per side on sA / sB with the exchange through the current stream).
            # Measured 0.624 vs 0.568 ms per Baby step: the executor then starts all three chains later.
            g_iu_, g_tu_, g_u0 = torch.empty_like(img_user), torch.empty_like(txt_user), torch.empty_like(img_user)
            g_ii_, g_ti_ = torch.empty_like(img_item), torch.empty_like(txt_item)
            rc = _lib.lib().mmssl_layer_combine_bwd2_f32(
                _ptr(img_user), _ptr(txt_user), _ptr(Gu), img_user.shape[0], _ptr(g_iu_), _ptr(g_tu_), _ptr(g_u0),
                _ptr(img_item), _ptr(txt_item), _ptr(Gi), img_item.shape[0], _ptr(g_ii_), _ptr(g_ti_), None,
                float(r), float(inv), _ptr(g_ss), 2.0, img_user.shape[1], _NORM_EPS, _lib.stream_ptr())
            _lib.check(rc, "mmssl_layer_combine_bwd2_f32")
            for st in (sA, sB):
                st.wait_stream(main)
            for t, st in ((g_ii_, sA), (g_iu_, sA), (g_ti_, sB), (g_tu_, sB)):
                if _os.environ.get("MMSSL_NO_RECORD_STREAM") != "1":
                    t.record_stream(st)
        elif split:
            with torch.cuda.stream(sA):
                g_iu_, g_tu_, g_u0 = _combine_bwd(img_user, txt_user, Gu, r, inv, g_ss, 2.0, True)
            with torch.cuda.stream(sB):
                g_ii_, g_ti_, _ = _combine_bwd(img_item, txt_item, Gi, r, inv, g_ss, 2.0, False)
            # each modal chain needs one tensor of the other side: exchange through the current stream
            # (a direct sA <-> sB event pair crashes hipGraph capture in this ROCm build)
            main.wait_stream(sA)
            main.wait_stream(sB)
            sA.wait_stream(main)
            sB.wait_stream(main)
        else:
            # gradients arrive on the modal outputs themselves (a caller used them elsewhere, e.g. u_sim -> D): both
            # combines and the adds on the current stream, then one fork
            g_iu_, g_tu_, g_u0 = _combine_bwd(img_user, txt_user, Gu, r, inv, g_ss, 2.0, True)
            g_ii_, g_ti_, _ = _combine_bwd(img_item, txt_item, Gi, r, inv, g_ss, 2.0, False)
            if G_img_item is not None:
                g_ii_ = g_ii_ + G_img_item
            if G_txt_item is not None:
                g_ti_ = g_ti_ + G_txt_item
            if G_img_user is not None:
                g_iu_ = g_iu_ + G_img_user
            if G_txt_user is not None:
                g_tu_ = g_tu_ + G_txt_user
            if overlap:
                for st in (sA, sB):
                    st.wait_stream(main)
                for t, st in ((g_ii_, sA), (g_iu_, sA), (g_ti_, sB), (g_tu_, sB)):
                    if _os.environ.get("MMSSL_NO_RECORD_STREAM") != "1":
                        t.record_stream(st)
        def chain_a1():
            with torch.cuda.stream(sA):
                out["g_x_img"] = _spmm_raw(ui, True, _spmm_raw(iu, True, g_ii_, EPI_AXPY, g_iu_, 1.0), EPI_NONE)

        def chain_a2():
            with torch.cuda.stream(sA):
                _, out["gW_img"], out["gb_img"] = _linear_wgrad_raw(out["g_x_img"], keep_img, scale, F_img, W_img)

        def chain_b1():
            with torch.cuda.stream(sB):
                out["g_x_txt"] = _spmm_raw(ui.twin(), True, _spmm_raw(iu.twin(), True, g_ti_, EPI_AXPY, g_tu_, 1.0), EPI_NONE)

        def chain_b2():
            with torch.cuda.stream(sB):
                _, out["gW_txt"], out["gb_txt"] = _linear_wgrad_raw(out["g_x_txt"], keep_txt, scale, F_txt, W_txt)

        def chain_a():
            chain_a1()
            chain_a2()

        def chain_b():
            chain_b1()
            chain_b2()

        # MMSSL_BWD_SCHED (experiments; needs the split form): 1 = the GCN chain starts when both modal SpMM pairs are
        # done, 2 = the weight-gradient GEMMs start when the GCN chain is done (the register-direct wgrad and the SpMMs
        # stretch each other when they run side by side), 0 = only the data dependencies
        sched = int(_os.environ.get("MMSSL_BWD_SCHED", "0")) if (split and not c_first) else 0
        if sched == 1:
            chain_a1()
            chain_b1()
            main.wait_stream(sA)
            main.wait_stream(sB)
            sC.wait_stream(main)
            chain_c()
            chain_a2()
            chain_b2()
        elif sched == 2:
            chain_c()
            chain_a1()
            chain_b1()
            main.wait_stream(sC)
            sA.wait_stream(main)
            sB.wait_stream(main)
            chain_a2()
            chain_b2()
        else:
            chains = {"A": chain_a, "B": chain_b, "C": chain_c}
            for c in _branch_order("MMSSL_BWD_ORDER", "CAB"):
                if not (c == "C" and c_first):
                    chains[c]()
        gi, gW_img, gb_img, gW_txt, gb_txt = out["gi"], out["gW_img"], out["gb_img"], out["gW_txt"], out["gb_txt"]
        if overlap:
            # g_u0 (sA before the exchange, or the current stream) and gi (sC) are what the embedding tables need; in
            # deferred mode the wgrad chains on sA / sB are left running (see defer_wgrad_join)
            for st in ((sC,) if (split and _DEFER["on"]) else (sA, sB, sC)):
                main.wait_stream(st)
        return (None, gW_img, gb_img if has_bi else None, None, None, gW_txt, gb_txt if has_bt else None, None, None,
                g_u0, gi, None, None, None, None, None)


def hot_forward(F_img, W_img, b_img, keep_img, F_txt, W_txt, b_txt, keep_txt, scale, u0, i0, ui, iu, n_layers, r,
                overlap=None):
    """(u_g, i_g, ss, img_item, txt_item, img_user, txt_user): see _HotForward."""
    if overlap is None:
        overlap = overlap_enabled()
    return _HotForward.apply(F_img, W_img, b_img, keep_img, F_txt, W_txt, b_txt, keep_txt, float(scale), u0, i0,
                             ui, iu, int(n_layers), float(r), bool(overlap))
