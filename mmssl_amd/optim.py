"""AdamW with the reference's update rule (torch.optim.AdamW as built at
/root/reference/MMSSL/main.py:76-80) as ONE kernel launch over all tensors that have a gradient
(`mmssl_adamw_f32`), with the step counter in device memory so the update can be replayed inside a
hipGraph. Drop-in for the subset of the torch optimizer API the trainer uses (`zero_grad`, `step`,
`param_groups`, `state_dict`/`load_state_dict` in torch's AdamW layout).

Restrictions against torch.optim.AdamW (documented, enforced where they can be): one step counter per param GROUP
(every parameter of a group must receive a gradient from the first step on, as all hot-path parameters do);
bias corrections are evaluated in fp32 on the device (torch's default path: host double) - the difference is
<= 1 ulp of the step size."""
import ctypes as _ct

import torch

from . import _lib

_MAX = 24        # MMSSL_ADAMW_MAX_TENSORS


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        self._steps = {}        # per param group: device float[2] {completed steps, block counter}
        self._zero_grads = {}   # parameter -> persistent all-zero gradient (parameters flagged `unused_grad_is_zero`)

    def _group_state(self, gi, device):
        st = self._steps.get(gi)
        if st is None:
            st = torch.zeros(2, dtype=torch.float32, device=device)
            self._steps[gi] = st
        return st

    @torch.no_grad()
    def fused_slots(self, weights, biases, pre_ticked=False):
        """What a kernel that applies this optimiser's update itself needs (ops.proj_wgrad(adam=...)): the tensors and
        moments of `weights` / `biases` (lists of parameters of ONE param group; bias entries may be None), the group's
        hyper-parameters and its device step counter. The caller excludes these parameters from step()."""
        gi = None
        for k, group in enumerate(self.param_groups):
            ids = {id(p) for p in group["params"]}
            if all(id(p) in ids for p in list(weights) + [b for b in biases if b is not None]):
                gi = k
                break
        if gi is None:
            raise _lib.MmsslError("FusedAdamW.fused_slots: the parameters must belong to one param group")
        group = self.param_groups[gi]

        def st(p):
            s = self.state[p]
            if not s:
                s["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                s["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            return s
        b1, b2 = group["betas"]
        return {"W": [p.data for p in weights], "mW": [st(p)["exp_avg"] for p in weights],
                "vW": [st(p)["exp_avg_sq"] for p in weights],
                "b": [None if p is None else p.data for p in biases],
                "mb": [None if p is None else st(p)["exp_avg"] for p in biases],
                "vb": [None if p is None else st(p)["exp_avg_sq"] for p in biases],
                "state": self._group_state(gi, weights[0].device), "lr": float(group["lr"]), "beta1": float(b1),
                "beta2": float(b2), "eps": float(group["eps"]), "weight_decay": float(group["weight_decay"]),
                "pre_ticked": bool(pre_ticked)}

    @torch.no_grad()
    def step(self, closure=None, groups=None, sliced=None, external_tick=False, exclude=None):
        """`groups`: optional iterable of param-group indices to update (each group has its own device step
        counter, so groups may be stepped at different points of one iteration).
        `sliced`: {parameter: (buffer, float offset, n_slices, stride)} - gradients that exist only as split-K partials
        (ops.take_wgrad_parts): slice s of the gradient starts at buffer[offset + s * stride]; the kernel adds the
        slices in order while it reads them. Such parameters need no `.grad`.
        `external_tick`: the caller already advanced this step's counter (step_counter) on the stream - a step object
        whose loss tail ticks every counter of the step in one launch - so no tick launch follows the update.
        `exclude`: parameters some other kernel of the step has already updated (fused_slots)."""
        skip = {id(p) for p in exclude} if exclude else ()
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if groups is not None and gi not in groups:
                continue
            todo = []
            for p in group["params"]:
                sl = sliced.get(p) if sliced else None
                if p.grad is None and sl is None and getattr(p, "unused_grad_is_zero", False) and id(p) not in skip:
                    # a parameter the forward reads nowhere but the reference's autograd hands an all-zero gradient
                    # (weight_dict.w_q, see Models.MMSSL.__init__): AdamW then applies its decoupled weight decay only
                    z = self._zero_grads.get(p)
                    if z is None:
                        z = self._zero_grads[p] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    p.grad = z
                if (p.grad is None and sl is None) or id(p) in skip:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise _lib.MmsslError("FusedAdamW: parameters must be contiguous fp32 HIP tensors")
                if p.grad is not None:
                    g = p.grad
                    if g.is_sparse:
                        raise _lib.MmsslError("FusedAdamW: sparse gradients are not supported")
                    if not g.is_contiguous():
                        g = g.contiguous()
                    gptr, nsl, gst = g.data_ptr(), 1, 0
                else:
                    buf, off, nsl, gst = sl
                    if buf.dtype != torch.float32 or not buf.is_contiguous() or off + (nsl - 1) * gst + p.numel() > buf.numel():
                        raise _lib.MmsslError("FusedAdamW: sliced gradient does not fit its buffer")
                    g, gptr = buf, buf.data_ptr() + 4 * off
                st = self.state[p]
                if not st:
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                todo.append((p, g, st["exp_avg"], st["exp_avg_sq"], gptr, nsl, gst))
                # this launch writes p through its raw pointer: torch's version counter does not see it, caches keyed on the
                # parameter's contents (ops.WeightPlanes) watch this serial
                p._mmssl_serial = getattr(p, "_mmssl_serial", 0) + 1
            if not todo:
                continue
            state = self._group_state(gi, todo[0][0].device)
            b1, b2 = group["betas"]
            # one launch per _MAX tensors; only the LAST launch of a group may advance the step counter,
            # so larger groups are not supported (the hot path has 7 tensors with gradients)
            if len(todo) > _MAX:
                raise _lib.MmsslError("FusedAdamW: more than %d tensors with gradients in one group" % _MAX)
            n = len(todo)
            arr = lambda k: (_ct.c_void_p * n)(*[t[k].data_ptr() for t in todo])       # noqa: E731
            numel = (_ct.c_int64 * n)(*[t[0].numel() for t in todo])
            gptrs = (_ct.c_void_p * n)(*[t[4] for t in todo])
            slices = (_ct.c_int32 * n)(*[t[5] for t in todo])
            gstride = (_ct.c_int64 * n)(*[t[6] for t in todo])
            rc = _lib.lib().mmssl_adamw_sliced_f32(arr(0), gptrs, arr(2), arr(3), numel, slices, gstride, n,
                                                   state.data_ptr(), float(group["lr"]), float(b1), float(b2),
                                                   float(group["eps"]), float(group["weight_decay"]),
                                                   1 if external_tick else 0, _lib.stream_ptr())
            _lib.check(rc, "mmssl_adamw_sliced_f32")
        return loss

    # torch.optim.AdamW-compatible checkpoint layout: per-parameter "step" tensors
    def state_dict(self):
        sd = super().state_dict()
        for gi, group in enumerate(sd["param_groups"]):
            st = self._steps.get(gi)
            for pid in group["params"]:
                if pid in sd["state"]:
                    sd["state"][pid]["step"] = (st[0:1].clone().reshape(()) if st is not None
                                                else torch.tensor(0.0))
        return sd

    @torch.no_grad()
    def reset_state(self):
        """Zero the moments and the step counters IN PLACE (addresses stay valid for a captured hipGraph)."""
        for st in self.state.values():
            for v in st.values():
                if torch.is_tensor(v):
                    v.zero_()
        for t in self._steps.values():
            t.zero_()

    def step_counter(self, gi, device):
        """Device float[2] {completed steps, -} of param group `gi` (created on first use)."""
        return self._group_state(gi, device)

    def load_state_dict(self, state_dict):
        """ONE step counter per param group (torch.optim.AdamW keeps one per parameter): a checkpoint whose
        parameters of one group disagree on `step` cannot be represented and is rejected."""
        steps = {}
        for gi, group in enumerate(state_dict["param_groups"]):
            for pid in group["params"]:
                st = state_dict["state"].get(pid)
                if st is not None and "step" in st:
                    if gi in steps and steps[gi] != float(st["step"]):
                        raise _lib.MmsslError("FusedAdamW: parameters of group %d have different step counts "
                                              "(%g vs %g); this optimiser keeps one counter per group" % (
                                                  gi, steps[gi], float(st["step"])))
                    steps[gi] = float(st["step"])
        super().load_state_dict(state_dict)
        for p_state in self.state.values():
            p_state.pop("step", None)
        for gi, val in steps.items():
            dev = self.param_groups[gi]["params"][0].device
            self._group_state(gi, dev)[0] = val
