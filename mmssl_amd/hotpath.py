"""The data-parallel hot path as ONE replayable unit: model forward -> BPR + InfoNCE x2 +
feature regulariser -> backward -> AdamW, i.e. the generator step of the reference loop
(/root/reference/MMSSL/main.py:363-429) without the adjacent adversarial pieces (Discriminator /
u_sim / gradient penalty: SURVEY.md section 2 rows 3 and 7, out of the hot-path scope).

All launches (hand-written HIP kernels through the C ABI + a few torch elementwise/optimizer
kernels) go to one stream and can be captured once into a hipGraph and replayed, which removes
the host launch gaps that dominate at these problem sizes (a Baby-shape SpMM is ~11 us).
"""
import torch

from . import ops
from .hotnode import HotCtx
from .optim import FusedAdamW
from .config import args


class HotPathStep:
    """One step = forward -> losses -> backward -> AdamW on this object's own stream and its own hotnode.HotCtx (side
    streams + the hand-offs between the forward node, the loss tail and the optimiser): nothing here is process-global,
    so several step objects (models, devices, host threads) can be built, captured and run side by side.

    Options (constructor arguments, fixed for the object's life; the defaults are the measured-best forms):
      overlap       fork the projection / modal chain and the GCN chain onto two side streams (False: one stream)
      eager_loss    root the backward at the loss TERMS with their known gradients, so that the loss section is one
                    chain of launches whose tail also assembles the loss and ticks the step's counters
                    (False: autograd through ops.loss_assemble, the op-by-op structure)
      batch_rows    (with eager_loss) the fused tables are computed at the batch's rows only before the loss chain - all
                    the loss reads - and the regulariser's |.|^2 sums run on the side stream next to that chain, joining
                    the loss value by one launch there in the backward: the dense fuse launch leaves the critical path
                    (False: the dense two-sided fuse kernel in front of the loss chain)
      fuse_adam     the projection weights' AdamW update is applied by the weight-gradient epilogue itself and — when the
                    modal graphs are empty, i.e. the embedding tables' gradients come from the hot node alone — every
                    other parameter is updated on the GCN chain's side stream while the weight gradient still runs: no
                    optimiser launch is left on the step's critical path (False: one launch after the backward)"""

    def __init__(self, model, graphs, batch_size, decay=1e-5, lr=None, capturable=True, overlap=True, eager_loss=True,
                 fuse_adam=True, batch_rows=True):
        self.model = model
        self.graphs = tuple(graphs)
        self.batch_size = int(batch_size)
        self.decay = float(decay)
        dev = model.user_id_embedding.weight.device
        self.batch = torch.zeros((3, batch_size), dtype=torch.int64, device=dev)     # users / pos / neg
        self.users, self.pos, self.neg = self.batch[0], self.batch[1], self.batch[2]
        # loss assembly weights for [mf, emb, reg, cl_img, cl_txt] (main.py:420 without the GAN term)
        self.loss_w = torch.tensor([1.0, 1.0, 1.0, args.cl_rate, args.cl_rate], dtype=torch.float32, device=dev)
        # same update rule as the reference's optim.AdamW (main.py:76-80) as ONE launch over all tensors
        # with a gradient; the step counter lives on the device so the step can be replayed in a hipGraph.
        # (`capturable` is kept for signature compatibility: the kernel always is.)
        self.optimizer = FusedAdamW([{"params": list(model.parameters())}], lr=lr or args.lr)
        self.loss = torch.zeros((), device=dev)
        self._one = torch.ones((), device=dev)
        self._feat_c, self._feat_c_val = None, None
        self.parts = {}
        self._graph = None
        self.eager_loss = bool(eager_loss)
        # Only the PACKED hot node (hotnode._HotNode: two modalities, a modal width the SpMM has, feature widths the
        # grouped projection takes) consumes the step's hand-offs: the fused AdamW slots of the projection weights, the
        # step-owned RNG tick of its mask draw, the batch-rows fuse. A model that MMSSL.forward routes through
        # _forward_multi (a third modality, d = 128, a 20-wide feature ...) gets none of them: every parameter is
        # updated by the optimiser launch and ops.dropout_masks ticks the generator itself.
        from .hotnode import packed_supported
        self._packed = (not getattr(model, "extra_names", None)
                        and packed_supported([model.image_feats.shape[1], model.text_feats.shape[1]], model.n_items,
                                             args.embed_size))
        self.fuse_adam = bool(fuse_adam) and self._packed
        self.batch_rows = bool(batch_rows) and self._packed
        self.hot = HotCtx(dev, overlap=overlap)
        self._proj = [model.image_trans, model.text_trans]
        if self.fuse_adam and ops.PROJ_SPLIT:
            self.hot.planes = ops.WeightPlanes()
        # tables first: with empty modal graphs their gradients are complete as soon as the GCN chain is
        self._modal_empty = (self._packed
                             and all(getattr(g, "nnz", 1) == 0 and not hasattr(g, "_pair") for g in self.graphs[2:6]))
        self._tables_early = self.fuse_adam and self.hot.overlap and self._modal_empty
        # parity runs inject fixed uint8 dropout keep-masks (img, txt), each [n_items, d]; None = drawn inside the
        # projection's epilogue (fresh masks on every replay)
        self.keep_masks = None
        self._ring = None
        # completed steps as a uint64 (the batch ring's slot index; the optimiser's own counter is an fp32 that stops
        # counting at 2^24): advanced by the loss tail's tick list together with the RNG launch counter
        self._steps_done = torch.zeros(1, dtype=torch.int64, device=dev)
        # ONE stream for everything this object launches (eager steps, capture, replays): autograd
        # binds each parameter's AccumulateGrad node to the stream of its first backward, and a
        # later capture on a different stream would have to synchronise across streams.
        # (Default priority. A high-priority step stream measured 5 us faster per step when it worked - and in about one run of
        # three the replayed graph then executed its branches one after the other: 0.76-0.83 ms per step, the sum of the
        # kernel times. profiles/r04/experiments_rejected.txt)
        self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))

    def set_batch(self, users, pos=None, neg=None):
        """Device-to-device copies into the static index buffers (graph replays read these)."""
        with torch.cuda.stream(self.stream):
            if pos is None:                       # packed [3, B] batch: one device-to-device copy
                self.batch.copy_(users, non_blocking=True)
            else:
                self.users.copy_(users, non_blocking=True)
                self.pos.copy_(pos, non_blocking=True)
                self.neg.copy_(neg, non_blocking=True)

    def set_batch_ring(self, batches):
        """`batches`: int64 [n, 3, B] device tensor of upcoming batches (users / pos / neg each). From now on every step
        copies slot (completed optimiser steps mod n) into its index buffers by a launch of its own on the GCN chain's
        side stream: replays need no set_batch() (no host-issued copy between two replays). None switches back."""
        if batches is None:
            self._ring = None
            return
        if batches.dtype != torch.int64 or batches.dim() != 3 or tuple(batches.shape[1:]) != (3, self.batch_size) \
                or batches.device != self.batch.device:
            raise ops._lib.MmsslError("set_batch_ring: int64 [n, 3, batch_size] tensor on the model's device expected")
        self._ring = batches.contiguous()

    def _select_batch(self):
        dev = self.loss.device
        rc = ops._lib.lib().mmssl_select_slot_i64(self._ring.data_ptr(), self._ring.shape[0], 3 * self.batch_size,
                                                  self._steps_done.data_ptr(), self.batch.data_ptr(),
                                                  ops._lib.stream_ptr())
        ops._lib.check(rc, "mmssl_select_slot_i64")

    def _feat_coeff(self):
        c = args.feat_reg_decay * 0.5 / self.model.n_items
        if self._feat_c is None or self._feat_c_val != c:
            self._feat_c, self._feat_c_val = torch.full((), c, dtype=torch.float32, device=self.loss.device), c
        return c

    def losses(self, ticks=None):
        """Forward + loss assembly through autograd-visible ops; returns (total, parts). The step's loss lands in the
        persistent buffer self.loss (read back by callers after a replay)."""
        m = self.model
        (ua, ia, img_item, txt_item, img_user, txt_user, uemb, _, img_uid, txt_uid, _, _) = m(
            *self.graphs, keep_masks=self.keep_masks, hot=self.hot)
        terms = ops.batch_losses_vec(ua, ia, img_uid, txt_uid, self.users, self.pos, self.neg, self.decay,
                                     self.batch_size, args.tau)                 # [mf, emb, 0, cl_img, cl_txt]
        ss = m.feat_sumsq(img_item, txt_item, img_user, txt_user)
        c = self._feat_coeff()
        # _step() backpropagates the persistent ones tensor: the assembly's gradients are the constants
        total = ops.loss_assemble(terms, self.loss_w, ss, c, out=self.loss, unit_grad_c=self._feat_c, ticks=ticks)
        return total, dict(terms=terms, ss=ss)

    def _losses_eager(self, ticks):
        """losses() for _step(): the terms' and the regulariser's gradients are known constants (loss_w, c), so the
        backward is rooted at them directly; the loss value (self.loss) and the step's counter ticks come out of the
        last launch of the loss section (ops._BatchLosses._forward_eager)."""
        m, hot = self.model, self.hot
        # batch-rows form: the forward fuses the batch's rows only and sums |.|^2 on the side stream (see __init__).
        # Otherwise it leaves its regulariser sum unreduced and the loss tail reduces it (one launch less in front of the
        # loss chain) - only valid because the very next consumer of `ss` IS that tail, which checks it.
        rows_mode = self.batch_rows
        hot.defer_ss, hot.ss_parts, hot.prefill_buf = (not rows_mode), None, None
        hot.reg_parts, hot.reg_target = None, None
        hot.batch_rows = (self.users, self.batch[1:3].reshape(-1)) if rows_mode else None
        # [g_ua | g_ia | g_img_uid | g_txt_uid | 3 tickets]: the modal id views only get gradients when they are not the
        # cached zeros of empty modal graphs
        n_views = 0 if self._modal_empty else 2
        hot.prefill_floats = lambda nu, ni, d: ((1 + n_views) * nu + ni) * d + 4
        try:
            (ua, ia, img_item, txt_item, img_user, txt_user, uemb, _, img_uid, txt_uid, _, _) = m(
                *self.graphs, keep_masks=self.keep_masks, hot=hot)
        finally:
            hot.defer_ss, hot.prefill_floats, hot.batch_rows = False, None, None
        ss = m.feat_sumsq(img_item, txt_item, img_user, txt_user)
        c = self._feat_coeff()
        rows_mode = hot.reg_parts is not None          # the packed node ran in batch-rows form
        terms = ops.batch_losses_vec(ua, ia, img_uid, txt_uid, self.users, self.pos, self.neg, self.decay,
                                     self.batch_size, args.tau, hot=hot, eager_w=self.loss_w,
                                     tail=((None if rows_mode else ss.detach()), c, self.loss, ticks))
        if rows_mode:
            hot.reg_target = (c, self.loss)            # the node's backward adds c * |.|^2 on its side stream
        self.parts = dict(terms=terms, ss=ss)
        return [terms, ss], [self.loss_w, self._feat_c]

    def step(self):
        """One eager step on this object's stream."""
        with torch.cuda.stream(self.stream):
            return self._step()

    def _step_multi(self):
        """The step of a model with further modalities (MMSSL(extra_feats=...), BASELINE configs[1]'s V/A/T): the loss over
        EVERY modality (one InfoNCE term and two regulariser tables per modality: oracle generator_loss_multi) out of the
        differentiable HIP ops, ordinary autograd, one AdamW launch over all parameters. None of the packed node's
        hand-offs apply here."""
        m = self.model
        self.optimizer.zero_grad(set_to_none=True)
        if self._ring is not None:          # no hot node to hang the slot pick on: select it in front of the forward
            self._select_batch()
        out = m(*self.graphs, keep_masks=self.keep_masks, extra_graphs=getattr(self, "extra_graphs", None))
        n_extra = (len(out) - 12) // 4
        mf, emb = ops.bpr_gather(out[0], out[1], self.users, self.pos, self.neg, self.decay, self.batch_size)
        feats = [2, 3, 4, 5] + [12 + 4 * k + j for k in range(n_extra) for j in (0, 1)]
        views = [8, 9] + [14 + 4 * k for k in range(n_extra)]
        ss = ops.sumsq(out[feats[0]])
        for k in feats[1:]:
            ss = ss + ops.sumsq(out[k])
        cl = ops.infonce(out[views[0]], out[0], args.tau, idx=self.users)
        for k in views[1:]:
            cl = cl + ops.infonce(out[k], out[0], args.tau, idx=self.users)
        total = mf + emb + (args.feat_reg_decay * 0.5 / m.n_items) * ss + args.cl_rate * cl
        total.backward()
        self.optimizer.step()
        self.loss.copy_(total.detach())
        self._steps_done.add_(1)             # the batch ring's slot index (the packed path's loss tail ticks it)
        return self.loss

    def _step(self):
        if getattr(self.model, "extra_names", None):
            return self._step_multi()
        self.optimizer.zero_grad(set_to_none=True)
        # the step owns its counters: the projection's mask draw and the AdamW launch run without their one-thread tick
        # kernels, the loss section's last launch (between them in stream order) advances the RNG launch counter and
        # the AdamW step counter
        dev = self.loss.device
        hot = self.hot
        counters = [self.optimizer.step_counter(gi, dev).data_ptr() for gi in range(len(self.optimizer.param_groups))]
        ticks = (counters, [self._steps_done.data_ptr()] + ([ops._rng_state(dev).data_ptr() + 8] if self._packed else []))
        hot.external_ticks, hot.lazy_anchors = True, True
        hot.side_prologue = self._select_batch if self._ring is not None else None
        fused = []
        if self.fuse_adam:
            fused = [l.weight for l in self._proj] + [l.bias for l in self._proj if l.bias is not None]
            hot.adam = self.optimizer.fused_slots([l.weight for l in self._proj], [l.bias for l in self._proj],
                                                  pre_ticked=True)
        try:
            if self.eager_loss:
                roots, grads = self._losses_eager(ticks)
                torch.autograd.backward(roots, grads)
            else:
                total, parts = self.losses(ticks)
                total.backward(gradient=self._one)       # persistent root gradient: no ones_like fill per step
            hot.assign_anchored_zero_grads()
            if self._tables_early and hot.after_fuse_bwd is not None:
                main = torch.cuda.current_stream(dev)
                sC = hot.streams()[1]
                sC.wait_event(hot.after_fuse_bwd)
                # the side stream may only run ahead of the current one if the tables' .grad ARE the node's buffers
                # (AccumulateGrad stole them); a cloned / accumulated gradient was written by a launch on the current
                # stream, which the update then has to wait for. Decided at capture time for a replayed graph.
                tg = hot.table_grads
                m = self.model
                if tg is None or any(p.grad is None or p.grad.data_ptr() != g
                                     for p, g in zip((m.user_id_embedding.weight, m.item_id_embedding.weight), tg)):
                    sC.wait_event(main.record_event())
                with torch.cuda.stream(sC):
                    self.optimizer.step(external_tick=True, exclude=fused)
                main.wait_stream(sC)
            else:
                self.optimizer.step(external_tick=True, exclude=fused)
        finally:
            hot.external_ticks, hot.lazy_anchors = False, False
            hot.anchored, hot.adam, hot.after_fuse_bwd, hot.side_prologue, hot.table_grads = [], None, None, None, None
        return self.loss

    # ---- hipGraph capture ---------------------------------------------------------------------
    def capture(self, warmup=3):
        """Warm up and capture one full step ON THE SAME side stream (autograd's AccumulateGrad
        nodes are bound to the stream of the first backward; capturing on another stream makes
        them synchronise across streams inside the capture, which the HIP runtime rejects).
        Returns True on success; the object stays usable in eager mode otherwise."""
        self.model.train()
        s = self.stream
        try:
            with torch.cuda.stream(s):
                for _ in range(warmup):
                    self._step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                self._step()
            torch.cuda.synchronize()
            self._graph = g
            return True
        except Exception as e:       # pragma: no cover - depends on the runtime
            self._graph = None
            self.capture_error = repr(e)
            torch.cuda.synchronize()
            return False

    def run(self):
        with torch.cuda.stream(self.stream):
            if self._graph is not None:
                pl = self.hot.planes
                if pl is not None and pl.key is not None:
                    # the captured forward reads the weights' bf16 planes made by the previous step's backward: if anything
                    # else has written the projection weights since (load_state_dict, a torch optimiser, copy_), remake them
                    ws = [l.weight for l in self._proj]
                    if pl.image_for(ws) is None:
                        pl.refresh(ws)
                self._graph.replay()
            else:
                self._step()


class SplitHotPath:
    """The hot path as TWO replayable segments for callers that do other work between the forward and the backward
    of one step — the reference loop does: `u_sim_calculation` and the discriminator consume the forward's modal
    outputs and their loss term sends gradients back into them (main.py:372-420).

        F  model forward -> BPR + 2x InfoNCE + feature regulariser -> hot part of main.py:420        (one hipGraph)
           ... caller: anything eager on the static outputs; writes d(extra loss)/d(modal outputs) into .extra_grads
        B  backward of hot loss (+ the extra output gradients) -> AdamW                               (one hipGraph)

    Both graphs are captured once (autograd runs at capture time only, like torch.cuda.make_graphed_callables); the
    parameters, optimiser state, step counters and the dropout generator are snapshotted around warm-up + capture,
    so capturing does not advance training. Valid as long as the six graph handles stay the same objects."""

    def __init__(self, model, graphs, optimizer, batch_size, decay, loss_w, feat_c):
        self.model, self.graphs, self.optimizer = model, tuple(graphs), optimizer
        self.batch_size, self.decay, self.feat_c = int(batch_size), float(decay), float(feat_c)
        dev = model.user_id_embedding.weight.device
        self.device = dev
        self.batch = torch.zeros((3, batch_size), dtype=torch.int64, device=dev)
        self.loss_w = torch.tensor(list(loss_w), dtype=torch.float32, device=dev)
        self.loss = torch.zeros((), device=dev)
        self._one = torch.ones((), device=dev)
        self._c = torch.full((), self.feat_c, dtype=torch.float32, device=dev)
        self.stream = torch.cuda.Stream(device=dev)
        self.outs = None            # the model's 12-tuple (static tensors of graph F)
        self.terms = None           # [mf, emb, 0, cl_img, cl_txt]
        self.extra_grads = None     # gradients w.r.t. outs[2..5] from the caller's loss terms
        self._gF = self._gB = None

    def _forward(self):
        m = self.model
        outs = m(*self.graphs)
        u, p, n = self.batch[0], self.batch[1], self.batch[2]
        terms = ops.batch_losses_vec(outs[0], outs[1], outs[8], outs[9], u, p, n, self.decay, self.batch_size, args.tau)
        ss = m.feat_sumsq(outs[2], outs[3], outs[4], outs[5])
        total = ops.loss_assemble(terms, self.loss_w, ss, self.feat_c, out=self.loss, unit_grad_c=self._c)
        return outs, terms, total

    def _backward(self, outs, total):
        self.optimizer.zero_grad(set_to_none=True)
        torch.autograd.backward([total] + [outs[k] for k in (2, 3, 4, 5)], [self._one] + list(self.extra_grads))
        self.optimizer.step()

    def _snapshot(self):
        st = {"p": [p.detach().clone() for p in self.model.parameters()],
              "rng": ops._rng_state(self.device).clone(),
              "opt": {id(p): {k: v.clone() for k, v in s.items() if torch.is_tensor(v)}
                      for p, s in self.optimizer.state.items()},
              "steps": {gi: t.clone() for gi, t in self.optimizer._steps.items()}}
        return st

    def _restore(self, st):
        with torch.no_grad():
            for p, q in zip(self.model.parameters(), st["p"]):
                p.copy_(q)
            ops._rng_state(self.device).copy_(st["rng"])
            for p, s in self.optimizer.state.items():
                old = st["opt"].get(id(p))
                for k, v in s.items():
                    if torch.is_tensor(v):
                        v.copy_(old[k]) if old is not None and k in old else v.zero_()
            for gi, t in self.optimizer._steps.items():
                t.copy_(st["steps"][gi]) if gi in st["steps"] else t.zero_()

    def capture(self, warmup=2):
        """Autograd binds a parameter's gradient accumulator to the stream on which the parameter first entered a
        graph, for as long as any such graph is alive. Eager steps before this call ran on another stream, so their
        graphs must be gone (callers keep detached losses only; gc below) or the captured backward would have to
        synchronise with that stream inside the capture - hipStreamEndCapture segfaults on that (ROCm 7.2)."""
        import gc
        # the model caches tensors of its latest forward (feature-regulariser sum, id views): they carry that
        # forward's graph - and through it the old accumulators, which every later forward would inherit
        self.model._feat_sumsq = None
        self.model.embedding_dict = {"user": {}, "item": {}}
        gc.collect()
        self.model.train()
        s = self.stream
        s.wait_stream(torch.cuda.current_stream(self.device))
        snap = self._snapshot()
        try:
            with torch.cuda.stream(s):
                for _ in range(warmup):
                    outs, terms, total = self._forward()
                    if self.extra_grads is None:
                        self.extra_grads = [torch.zeros_like(outs[k]) for k in (2, 3, 4, 5)]
                    self._backward(outs, total)
                    del outs, terms, total          # nothing of the warm-up is released inside the capture region
            torch.cuda.synchronize()
            gF, gB = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(gF, stream=s, capture_error_mode="thread_local"):
                outs, terms, total = self._forward()
            with torch.cuda.graph(gB, stream=s, pool=gF.pool(), capture_error_mode="thread_local"):
                self._backward(outs, total)
            torch.cuda.synchronize()
            self.outs, self.terms, self._gF, self._gB = outs, terms, gF, gB
            ok = True
        except Exception as e:       # pragma: no cover - depends on the runtime
            self.capture_error, ok = repr(e), False
            self._gF = self._gB = None
            torch.cuda.synchronize()
        self._restore(snap)
        torch.cuda.current_stream(self.device).wait_stream(s)
        return ok

    def forward(self, batch3):
        """Replay F on the packed [3, B] int64 batch; returns the model's output tuple (static tensors: valid until
        the next replay). The caller's stream is ordered after the replay."""
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self.batch.copy_(batch3, non_blocking=True)
            self._gF.replay()
        cur.wait_stream(self.stream)
        return self.outs

    def backward(self, extra=None):
        """Replay B. `extra`: four tensors d(caller loss)/d(outs[2..5]) or None (zeros)."""
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            for dst, src in zip(self.extra_grads, extra or (None,) * 4):
                dst.zero_() if src is None else dst.copy_(src, non_blocking=True)
            self._gB.replay()
        cur.wait_stream(self.stream)
