"""Peer exchange: the transport of mmssl_amd/dist.py's table exchanges WITHOUT a collective library in the data path
(csrc/peer.hip; C ABI `mmssl_peer_*` in include/mmssl_hip.h). One process per GPU as before; torch.distributed is used
once per window for the HOST-side exchange of IPC handles, never for tensor data.

    all-gather       every rank's kernel pushes its rows into the same row range of EVERY rank's window (all xGMI links
                     at once, no ring), its last block publishes the channel's epoch; the consumer's stream waits for all
                     epochs (the push's last block does, after the others have left). ONE launch, no staging copy: the
                     window IS the gathered table.
    reduce-scatter   the partial products are written straight into the rank's window (`partial()` hands it to the SpMM as
                     its output), signal + wait (one wave, one launch), then ONE kernel pulls this rank's rows out of every
                     window and adds them in rank order - bit-reproducible, the same bits on every rank.
    all-reduce       (small replicated buffers) push into slot [rank] of every window + local fixed-order sum.

Windows are per CALL SITE: the k-th exchange of a step always uses window k (all ranks run the same sequence of calls), so
within a step no window is written twice; `begin_step()` is a barrier on channel 0 in front of a step's first exchange,
after which every window may be overwritten again (each rank has finished the previous step, i.e. every read of every
window it was handed). Pointers and channels are fixed after the first (warm-up) step: the kernels are captured into the
step's hipGraph like any other launch.

The reference has no multi-device path (MMSSL/main.py:529); BASELINE.json's north_star names the all-gather of neighbour
embeddings before each propagation layer (Models.py:201-211) - `dist._TableExchange` over RCCL stays as the A/B."""
import ctypes

import torch
import torch.distributed as dist

from . import _lib

_ptr = lambda t: ctypes.c_void_p(t.data_ptr())        # noqa: E731


class _DevMem:
    """A device buffer owned by the C library, exposed to torch through __cuda_array_interface__."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(int(s) for s in shape), "typestr": "<f4", "data": (int(ptr), False),
                                         "version": 2, "strides": None}


class PeerTransport:
    """The context of one rank: flags + windows. Creation and `window()` are host-side collectives over `group`."""

    def __init__(self, group, device, max_channels=1024, timeout_ms=20000):
        self.group, self.device = group, torch.device(device)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        L = _lib.lib()
        self._L = L
        self.hb = L.mmssl_peer_handle_bytes()
        self.max_channels = int(max_channels)
        self._keep = []
        self.launches = 0
        self._ctx = None

        def local():
            ctx = ctypes.c_void_p()
            _lib.check(L.mmssl_peer_create(self.world, self.rank, int(max_channels), ctypes.byref(ctx)), "mmssl_peer_create")
            self._ctx = ctx
            _lib.check(L.mmssl_peer_set_timeout_ms(ctx, int(timeout_ms)), "mmssl_peer_set_timeout_ms")
            h = ctypes.create_string_buffer(self.hb)
            _lib.check(L.mmssl_peer_flags_handle(ctx, h), "mmssl_peer_flags_handle")
            return h.raw
        with torch.cuda.device(self.device):
            hs = self._all_handles(local)
            if self.world > 1:
                self._agree(lambda: _lib.check(L.mmssl_peer_open_flags(self._ctx, hs), "mmssl_peer_open_flags"))

    # Set-up is a sequence of LOCAL steps (allocate + export, open the peers' handles) and host-side exchanges. A local
    # step that fails on one rank only must not leave the other ranks alone in the next exchange: the failure travels
    # through the exchange itself and every rank raises the same error at the same point.
    def _all_handles(self, make):
        try:
            mine = bytes(make())
        except Exception as e:                    # noqa: BLE001 - reported to every rank below
            mine = "rank %d: %s" % (self.rank, repr(e)[:200])
        got = [None] * self.world
        dist.all_gather_object(got, mine, group=self.group)
        bad = [g for g in got if not (isinstance(g, (bytes, bytearray)) and len(g) == self.hb)]
        if bad:
            raise _lib.MmsslError("peer exchange set-up failed (%s)" % (bad[0],))
        return ctypes.create_string_buffer(b"".join(got), self.hb * self.world)

    def _agree(self, step):
        try:
            step()
            mine = None
        except Exception as e:                    # noqa: BLE001
            mine = "rank %d: %s" % (self.rank, repr(e)[:200])
        got = [None] * self.world
        dist.all_gather_object(got, mine, group=self.group)
        bad = [g for g in got if g is not None]
        if bad:
            raise _lib.MmsslError("peer exchange set-up failed (%s)" % (bad[0],))

    def info(self):
        v = (ctypes.c_int64 * 6)()
        _lib.check(self._L.mmssl_peer_info(self._ctx, v), "mmssl_peer_info")
        return {"world": v[0], "rank": v[1], "channels": v[2], "flags_finegrained": bool(v[3]), "windows": v[4],
                "window_bytes": v[5]}

    def window(self, rows, width):
        """(window id, torch view [rows, width] of this rank's buffer). Collective."""
        wid, ptr = ctypes.c_int(), ctypes.c_void_p()

        def local():
            h = ctypes.create_string_buffer(self.hb)
            _lib.check(self._L.mmssl_peer_window_create(self._ctx, int(rows) * int(width) * 4, ctypes.byref(wid), h,
                                                        ctypes.byref(ptr)), "mmssl_peer_window_create")
            return h.raw
        with torch.cuda.device(self.device):
            if self.world > 1:
                hs = self._all_handles(local)
                self._agree(lambda: _lib.check(self._L.mmssl_peer_window_open(self._ctx, wid.value, hs),
                                               "mmssl_peer_window_open"))
            else:
                local()
        t = torch.as_tensor(_DevMem(ptr.value, (rows, width)), device=self.device)
        assert t.data_ptr() == ptr.value
        return wid.value, t

    # ---- stream-ordered launches on torch's current stream ----
    def push_rows(self, ch, wid, src, row0, dst_pitch, wait=False):
        """wait=True: the launch's last block also waits for every peer's push on the channel (no separate wait launch)."""
        assert src.dim() == 2 and src.stride(1) == 1 and src.dtype == torch.float32
        self.launches += 1
        _lib.check(self._L.mmssl_peer_push_rows_f32(self._ctx, ch, wid, _ptr(src), src.stride(0) if src.shape[0] > 1 else
                                                    src.shape[1], src.shape[0], src.shape[1], int(row0), int(dst_pitch),
                                                    1 if wait else 0, _lib.stream_ptr()), "mmssl_peer_push_rows_f32")

    def signal_wait(self, ch):
        self.launches += 1
        _lib.check(self._L.mmssl_peer_signal_wait(self._ctx, ch, _lib.stream_ptr()), "mmssl_peer_signal_wait")

    def signal(self, ch):
        self.launches += 1
        _lib.check(self._L.mmssl_peer_signal(self._ctx, ch, _lib.stream_ptr()), "mmssl_peer_signal")

    def wait(self, ch):
        self.launches += 1
        _lib.check(self._L.mmssl_peer_wait(self._ctx, ch, _lib.stream_ptr()), "mmssl_peer_wait")

    def pull_sum(self, wid, row0, rows, width, pitch, out):
        self.launches += 1
        _lib.check(self._L.mmssl_peer_pull_sum_rows_f32(self._ctx, wid, int(row0), int(rows), int(width), int(pitch),
                                                        _ptr(out), out.stride(0) if out.shape[0] > 1 else out.shape[1],
                                                        _lib.stream_ptr()), "mmssl_peer_pull_sum_rows_f32")

    def sum_slots(self, slots, n, stride, length, out):
        self.launches += 1
        _lib.check(self._L.mmssl_peer_sum_slots_f32(_ptr(slots), int(n), int(stride), int(length), _ptr(out),
                                                    _lib.stream_ptr()), "mmssl_peer_sum_slots_f32")

    def check(self):
        """Host-blocking: raises if any wait of this rank gave up (a peer died or fell > timeout behind)."""
        e = ctypes.c_uint32()
        with torch.cuda.device(self.device):
            _lib.check(self._L.mmssl_peer_error(self._ctx, ctypes.byref(e)), "mmssl_peer_error")
        if e.value:
            raise _lib.MmsslError("peer exchange: a wait on rank %d timed out (peer mask 0x%x): results of this step are "
                                  "invalid" % (self.rank, e.value))

    def close(self):
        if getattr(self, "_ctx", None):
            with torch.cuda.device(self.device):
                self._L.mmssl_peer_destroy(self._ctx)
            self._ctx = None


class PeerComm:
    """The exchanges of one process group over a PeerTransport, addressed by call order (see the module docstring)."""

    def __init__(self, group, device, timeout_ms=20000):
        self.t = PeerTransport(group, device, timeout_ms=timeout_ms)
        self.world, self.rank = self.t.world, self.t.rank
        self.k = 0                       # exchanges issued since begin_step()
        self.kp = 0                      # partial() buffers handed out since begin_step()
        self._next_ch = 1                # channel 0 = the step barrier
        self.slots = {}                  # (k, kind, rows, width) -> (channel, window id, window tensor)
        self.partials = {}               # (kp, rows, width) -> (channel, window id, window tensor)
        self._by_ptr = {}                # data_ptr of a partial window -> its slot
        self.log = None                  # dist.COMM["log"]-style accounting, set by dist

    def _channel(self):
        ch = self._next_ch
        self._next_ch += 1
        if ch >= self.t.max_channels:
            raise _lib.MmsslError("peer exchange: more than %d exchange call sites per step" % self.t.max_channels)
        return ch

    def begin_step(self):
        """Barrier in front of a step's first exchange: every rank has finished the previous step."""
        self.t.signal_wait(0)
        self.k = self.kp = 0

    def _slot(self, kind, rows, width):
        key = (self.k, kind, int(rows), int(width))
        self.k += 1
        s = self.slots.get(key)
        if s is None:
            wid, t = self.t.window(rows, width)
            s = self.slots[key] = (self._channel(), wid, t)
        return s

    # ---- all-gather: [per, w] row shard (row-pitched views allowed) -> [world * per, w], the window itself ----
    def gather(self, x):
        per, w = x.shape
        ch, wid, win = self._slot("g", self.world * per, w)
        self.t.push_rows(ch, wid, x, self.rank * per, w, wait=True)
        return win

    # ---- reduce-scatter: [world * per, w] partial products -> this rank's [per, w] rows of their sum ----
    def partial(self, rows, width, device=None):
        """The buffer the NEXT partial product of this shape should be written into (then passed to `reduce`)."""
        key = (self.kp, int(rows), int(width))
        self.kp += 1
        s = self.partials.get(key)
        if s is None:
            wid, t = self.t.window(rows, width)
            s = self.partials[key] = (self._channel(), wid, t)
            self._by_ptr[t.data_ptr()] = s
        return s[2]

    def reduce(self, P, per):
        rows, w = P.shape
        s = self._by_ptr.get(P.data_ptr()) if P.is_contiguous() else None
        if s is None or tuple(s[2].shape) != (rows, w):        # a partial that was not produced in place: one copy
            ch, wid, win = self._slot("r", rows, w)
            win.copy_(P)
        else:
            ch, wid, win = s
        self.t.signal_wait(ch)
        out = torch.empty((per, w), dtype=torch.float32, device=P.device)
        self.t.pull_sum(wid, self.rank * per, per, w, w, out)
        return out

    # ---- all-reduce (sum) of a small fp32 buffer, in place ----
    def all_reduce_(self, t):
        flat = t.view(-1) if t.is_contiguous() else None
        if flat is None or t.dtype != torch.float32:
            raise _lib.MmsslError("peer all-reduce: contiguous fp32 buffers only")
        n = flat.numel()
        n4 = (n + 3) // 4 * 4
        ch, wid, win = self._slot("a", self.world + 1, n4)          # slots [0, world) + one staging row
        src = flat.view(1, n)
        if n4 != n or (flat.data_ptr() & 15):
            stage = win[self.world]
            stage[:n].copy_(flat)
            src = stage.view(1, n4)
        self.t.push_rows(ch, wid, src, self.rank, n4, wait=True)
        self.t.sum_slots(win, self.world, n4, n, flat)
        return t

    def stats(self):
        i = self.t.info()
        i.update(call_sites=len(self.slots) + len(self.partials), launches=self.t.launches)
        return i

    def check(self):
        self.t.check()

    def close(self):
        self.t.close()
