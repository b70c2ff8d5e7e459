"""MMSSL.forward after the id-embedding fusion as ONE autograd node over PACKED modal features
(/root/reference/MMSSL/Models.py:173-174, 177-178, 182-183, 199-218):

    X   [I, nm d] = dropout(F_m W_m^T + b_m), m = image, text (, ...)   ONE grouped stream-K launch (csrc/projection.hip)
    MU  [U, nm d] = A_ui . X          MI [I, nm d] = A_iu . MU          two SpMMs of width nm d instead of 2 nm of width d:
                                                                        the edge lists are read once for all modalities
    u_l = A_ui . i_{l-1}, i_l = A_iu . u_l (softmax on the last layer)   the G-layer GCN chain, next to the modal chain
    u_g = mean_l(u_l) + r sum_m normalize(MU_m)   (items alike)          ONE fuse launch for both sides, which also leaves
    ss  = sum |MU|^2 + |MI|^2                                            the feature regulariser's sum of squares

Two chains are independent and bound by different resources - projection + modal SpMMs (fp32 MFMA, then gather) and the
GCN chain (gather, latency). The first is the step's critical path and stays on the current stream; the GCN chain is
forked onto a side stream and joined before the fuse kernel; the backward mirrors that (GCN backward on the side stream
|| fuse backward -> modal SpMMs -> grouped weight gradient on the current one).

Everything a step object wants to change about the node (deferred regulariser sum, the loss tail's pre-filled buffer,
step-owned counter ticks, lazy zero gradients) lives on a `HotCtx` owned by
that step object: there is no module-level state, two step objects on two streams do not see each other.
"""
import torch

from . import _lib, ops


class HotCtx:
    """Streams and hand-offs of one user of the hot node (a model called on its own, or one step object)."""

    def __init__(self, device, overlap=True):
        self.device = torch.device(device)
        self.overlap = bool(overlap) and self.device.type == "cuda"
        self._streams = None
        # --- set by a step object that owns the whole step (hotpath.HotPathStep); all off for a plain model(...) call
        self.defer_ss = False         # the forward leaves `ss` UNREDUCED: (ss, partials) wait in .ss_parts for the loss tail
        self.ss_parts = None
        self.prefill_floats = None    # fn(n_users, n_items, d) -> size of the loss tail's zero-filled buffer
        self.prefill_buf = None       # filled on the GCN chain's idle stream by the forward, consumed by the loss tail
        self.external_ticks = False   # the step's loss tail advances the RNG / AdamW counters: no tick launches here
        self.lazy_anchors = False     # zero gradients of skipped branches: assigned after the backward, no fill launch
        self.adam = None              # optim.FusedAdamW.fused_slots of the projection weights: the weight-gradient
        #                               epilogue applies their AdamW update itself (the caller excludes them from step())
        self.side_prologue = None     # callable run first on the GCN chain's side stream in the forward (a step object's
        #                               small launches that must precede the loss section but not the projection)
        self.after_fuse_bwd = None    # event on the current stream after the backward's fuse kernel: from there on the
        #                               gradient of u_0 exists (the GCN chain's side stream holds the one of i_0)
        self.table_grads = None       # (g_u0, g_i0) as the node's backward returned them: a step object that updates the
        #                               tables on the side stream checks that autograd STOLE these buffers as .grad
        #                               (a cloned gradient would be written by a copy on the current stream)
        self.batch_rows = None        # (user rows [B], item rows [2B]) int64 device tensors: the ONLY rows of the fused
        #                               tables the caller will read (a training step's loss). The forward then computes
        #                               just those rows (the other rows of u_g / i_g stay UNDEFINED); the regulariser's
        #                               |Mod|^2 sums come out of the backward's fuse kernel and join the loss value by a
        #                               launch on the side stream there (reg_target = (c, total buffer) names where).
        self.reg_parts = None         # (None, ss tensor) left by such a forward: its backward owes the regulariser
        self.reg_target = None
        self.planes = None            # ops.WeightPlanes of a step object that updates the projection weights in the weight
        #                               gradient's epilogue (adam above): the weights' bf16 planes are made right behind that
        #                               update, and the next forward starts with its main kernel instead of the split launch
        self.anchored = []
        self._zero_grads = {}

    def streams(self):
        """(auxiliary stream, GCN chain stream), created on first use."""
        if self._streams is None:
            self._streams = (torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device))
        return self._streams

    def join(self):
        """Make the current stream wait for everything queued on this context's side streams."""
        if self._streams is not None:
            main = torch.cuda.current_stream(self.device)
            for st in self._streams:
                main.wait_stream(st)

    def assign_anchored_zero_grads(self):
        """Give every parameter anchored since the last call an exactly-zero gradient if autograd produced none."""
        params, self.anchored = self.anchored, []
        for w in params:
            if w.grad is None:
                key = (w.data_ptr(), tuple(w.shape))
                z = self._zero_grads.get(key)
                if z is None:
                    z = torch.zeros_like(w)
                    self._zero_grads[key] = z
                w.grad = z


def packed_supported(feat_dims, n_items, d):
    """True when the packed node runs this modality list: the grouped projection takes it and the modal SpMM width
    nm * d is one the SpMM kernel has."""
    nm = len(feat_dims)
    return (1 <= nm <= 4 and (nm * d) in (32, 64, 128, 256) and d in (32, 64, 128, 256)
            and ops.proj_supported(feat_dims, n_items, d) and ops.proj_supported(feat_dims, n_items, d, wgrad=True))


class _HotNode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hot, nm, scale, p_drop, keep, ui, iu, n_layers, r, u0, i0, *flat):
        Fs, Ws, bs = flat[:nm], flat[nm:2 * nm], flat[2 * nm:3 * nm]
        u0, i0 = ops._chk(u0, "u0"), ops._chk(i0, "i0")
        dev = u0.device
        d = u0.shape[1]
        # The projection -> modal SpMMs -> fuse chain IS the step's critical path (the GCN chain is shorter and
        # independent of the projection weights): it stays on the CURRENT stream, so that no cross-queue edge of a
        # replayed hipGraph (10-15 us of wake-up latency each) lies on it; only the GCN chain is forked.
        main = torch.cuda.current_stream(dev)
        sC = hot.streams()[1] if hot.overlap else main
        # The replayed graph's executor keeps a node's FIRST-recorded successor on the node's queue and forks the others
        # (wake-up latency): the fork point is marked by an event, the critical chain's launch is recorded first, the
        # side stream then waits for the event (not for that launch).
        fork = main.record_event() if hot.overlap else None
        draw = None
        if keep is None and p_drop > 0.0:
            draw = (p_drop, ops._rng_state(dev))
        # with the GCN chain forked beside it the projection asks for 13/16 of the CUs (ops.proj_step_blocks)
        pblocks = ops.proj_step_blocks(dev) if hot.overlap else 0
        X, keep_used = ops.proj_forward(Fs, Ws, bs, keep=keep, draw=draw, scale=scale, blocks=pblocks, planes=hot.planes)
        if draw is not None and not hot.external_ticks:
            ops.tick_rng(dev)
        if hot.overlap:
            sC.wait_event(fork)
            u0.record_stream(sC)
            i0.record_stream(sC)
        # a third stream for what only the loss section needs (the step's batch selection, the zero-filled gradient
        # buffer): off the projection's AND off the GCN chain's path
        sA = hot.streams()[0] if hot.overlap else main
        if hot.overlap and (hot.side_prologue is not None or hot.prefill_floats is not None):
            sA.wait_event(fork)
        with torch.cuda.stream(sA):
            if hot.side_prologue is not None:
                hot.side_prologue()
            if hot.prefill_floats is not None:
                buf = torch.zeros(hot.prefill_floats(u0.shape[0], i0.shape[0], d), dtype=torch.float32, device=dev)
                if hot.overlap:
                    buf.record_stream(main)
                hot.prefill_buf = buf
        with torch.cuda.stream(sC):
            us, its = [u0], [i0]
            u, i = u0, i0
            uic, iuc = ui.twin(2), iu.twin(2)
            for l in range(n_layers):
                epi = ops.EPI_SOFTMAX if l == n_layers - 1 else ops.EPI_NONE
                u = ops._spmm_raw(uic, False, i, epi)
                i = ops._spmm_raw(iuc, False, u, epi)
                us.append(u)
                its.append(i)
        MU = ops._spmm_raw(ui, False, X, ops.EPI_NONE)
        MI = ops._spmm_raw(iu, False, MU, ops.EPI_NONE)
        if hot.overlap:
            main.wait_stream(sC)
            if hot.side_prologue is not None or hot.prefill_floats is not None:
                main.wait_stream(sA)
            for t in tuple(us[1:]) + tuple(its[1:]):
                t.record_stream(main)
        inv = 1.0 / (n_layers + 1)
        ss = torch.empty((), dtype=torch.float32, device=dev)
        if hot.batch_rows is not None:
            # the loss reads the fused tables at its batch rows only: those rows now (one small launch, the same arithmetic
            # as the dense kernel); the |Mod|^2 sums of the regulariser come out of the backward's fuse kernel and join the
            # loss value on the side stream
            u_g, i_g = ops.fuse_fwd_rows([(us, MU, hot.batch_rows[0]), (its, MI, hot.batch_rows[1])], inv, nm, r)
            hot.reg_parts = (None, ss)         # the backward's fuse kernel leaves the |Mod|^2 partials (it needs the norms)
        else:
            nbu, nbi = ops.fuse_blocks(u0.shape[0], d, nm), ops.fuse_blocks(i0.shape[0], d, nm)
            part = torch.empty(nbu + nbi, dtype=torch.float32, device=dev)
            u_g, i_g = ops.fuse_fwd([(us, MU, part[:nbu]), (its, MI, part[nbu:])], inv, nm, r)   # both sides, one launch
            if hot.defer_ss:
                hot.ss_parts = (ss, part)  # the caller's loss tail reduces the partials and stores the sum into `ss`
            else:
                rc = _lib.lib().mmssl_sum_partials_f32(part.data_ptr(), nbu + nbi, ss.data_ptr(), _lib.stream_ptr())
                _lib.check(rc, "mmssl_sum_partials_f32")
        ctx.save_for_backward(MU, MI, us[-1], its[-1], keep_used, *Fs)
        ctx.cfg = (hot, nm, float(scale), ui, iu, n_layers, float(r), inv, [b is not None for b in bs])
        ctx.proj_weights = list(Ws)       # the Parameter objects (not saved tensors: only identity and version are read)
        ctx.set_materialize_grads(False)
        return u_g, i_g, ss, MI, MU

    @staticmethod
    def backward(ctx, Gu, Gi, g_ss, G_MI, G_MU):
        MU, MI, uG, iG, keep = ctx.saved_tensors[:5]
        Fs = ctx.saved_tensors[5:]
        hot, nm, scale, ui, iu, n_layers, r, inv, has_b = ctx.cfg
        Gu = ops._chk(Gu, "Gu") if Gu is not None else torch.zeros_like(uG)
        Gi = ops._chk(Gi, "Gi") if Gi is not None else torch.zeros_like(iG)
        g_ss = g_ss.contiguous().to(torch.float32) if g_ss is not None else None
        G_MI = ops._chk(G_MI, "G_MI") if G_MI is not None else None
        G_MU = ops._chk(G_MU, "G_MU") if G_MU is not None else None
        dev = Gu.device
        d = Gu.shape[1]
        main = torch.cuda.current_stream(dev)
        sC = hot.streams()[1] if hot.overlap else main
        fork = main.record_event() if hot.overlap else None        # see forward: the critical launch is recorded first
        # critical chain (current stream): normalise-backward + regulariser gradient of both sides in one launch (the user
        # side also yields the gradient of u_0), the two modal SpMMs, the grouped weight gradient
        reg = hot.reg_parts
        part = None
        if reg is not None:
            nbu, nbi = ops.fuse_blocks(MU.shape[0], d, nm), ops.fuse_blocks(MI.shape[0], d, nm)
            part = torch.empty(nbu + nbi, dtype=torch.float32, device=dev)
        (gMU, g_u0), (gMI, _) = ops.fuse_bwd([(MU, Gu, G_MU, True), (MI, Gi, G_MI, False)], nm, r, inv, g_ss, 2.0,
                                             sumsq_part=None if part is None else [part[:nbu], part[nbu:]])
        if hot.overlap:
            hot.after_fuse_bwd = main.record_event()
            g_u0.record_stream(sC)             # a step object may run the tables' optimiser launch on the side stream
            sC.wait_event(fork)
            for t_ in (uG, iG, Gu, Gi):
                t_.record_stream(sC)           # main-pool tensors read on the side stream, possibly after this returns
        # g(MU) = A_iu^T g(MI) + own branch (AXPY epilogue). Recorded BEFORE anything on the side stream that depends on the
        # fuse kernel: the replayed graph keeps a node's first-recorded dependent on the node's queue, and that must be
        # this critical launch, not the regulariser's small add below
        t = ops._spmm_raw(iu, True, gMI, ops.EPI_AXPY, gMU, 1.0)
        # GCN chain (side stream): needs Gu / Gi only. last layer: i_G only feeds the mean; u_G feeds the mean and A_iu.u_G
        with torch.cuda.stream(sC):
            uic, iuc = ui.twin(2), iu.twin(2)
            gi = ops.softmax_rows_bwd(iG, Gi, inv)
            gu = ops._spmm_raw(iuc, True, gi, ops.EPI_AXPY_SOFTMAX_BWD, Gu, inv, uG)
            gi = ops._spmm_raw(uic, True, gu, ops.EPI_AXPY, Gi, inv)
            for _ in range(n_layers - 1):
                gu = ops._spmm_raw(iuc, True, gi, ops.EPI_AXPY, Gu, inv)
                gi = ops._spmm_raw(uic, True, gu, ops.EPI_AXPY, Gi, inv)
            if reg is not None:
                # the regulariser joins the loss value: total += c * sum |Mod|^2 from the fuse kernel's partials, behind
                # the GCN chain on its stream (nothing waits for the loss value inside the step)
                if hot.reg_target is None:
                    raise _lib.MmsslError("hot node: a batch-rows forward needs reg_target = (c, total) before its backward")
                if hot.overlap:
                    sC.wait_event(hot.after_fuse_bwd)
                    part.record_stream(sC)
                    reg[1].record_stream(sC)
                ops.loss_add_partials(part, hot.reg_target[0], hot.reg_target[1], reg[1])
                hot.reg_parts, hot.reg_target = None, None
        # g(X) = dropout-backward(A_ui^T g(MU)) (mask epilogue)
        if keep is not None:
            gX = ops.spmm_mask_raw(ui, True, t, keep, d, scale)
        else:
            gX = ops._spmm_raw(ui, True, t, ops.EPI_NONE)
        # the epilogue writes the updated weights AND (given a weight-planes image of the right shape) their bf16 planes for
        # the NEXT forward, which then starts with the projection's main kernel; without an image yet (first step) the planes
        # are made by a split launch behind the update
        fused_planes = hot.planes is not None and hot.adam is not None and hot.planes.writable_for(ctx.proj_weights)
        gW, gb = ops.proj_wgrad(gX, Fs, want_bias=any(has_b), adam=hot.adam,
                                blocks=ops.proj_step_blocks(gX.device) if hot.overlap else 0,
                                planes=hot.planes if fused_planes else None)
        if hot.planes is not None and hot.adam is not None:
            if fused_planes:
                hot.planes.mark_current(ctx.proj_weights)
            else:
                hot.planes.refresh(ctx.proj_weights)
        if hot.overlap:
            # the embedding-table gradient (GCN chain) is complete before anything downstream of this node runs: the
            # chain ends long before the weight gradient above does, so the join never waits
            main.wait_stream(sC)
            gi.record_stream(main)
        grads_b = [(gb[k] if (gb is not None and has_b[k]) else None) for k in range(nm)]
        hot.table_grads = (g_u0.data_ptr(), gi.data_ptr())     # addresses only: a reference would keep AccumulateGrad from stealing the buffers
        return (None,) * 9 + (g_u0, gi) + (None,) * nm + tuple(gW) + tuple(grads_b)


def hot_node(hot, Fs, Ws, bs, keep, p_drop, scale, u0, i0, ui, iu, n_layers, r):
    """(u_g, i_g, ss, MI [I, nm d], MU [U, nm d]); see the module docstring. `keep`: given uint8 [nm, I, d] masks, or
    None with p_drop > 0: the masks are drawn inside the projection's epilogue."""
    nm = len(Fs)
    Fs = [ops._chk(f, "F") for f in Fs]
    return _HotNode.apply(hot, nm, float(scale), float(p_drop), keep, ui, iu, int(n_layers), float(r), u0, i0,
                          *Fs, *Ws, *bs)
