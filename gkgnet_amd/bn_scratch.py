"""Per-device fp64 column-sum scratch of the two-launch BN passes (gkg_bn_apply_train / gkg_bn_bwd_atomic): the host-side
bookkeeping of which of the two alternating buffers is clean."""
from __future__ import annotations

import contextlib

import torch

from . import _lib
from .ops import _ptr, _stream      # noqa: F401


class _BnBwdScratch:
    """Two fp64 column-sum buffers per device for gkg_bn_apply_train / gkg_bn_bwd_atomic, used alternately: a call accumulates into the clean one
    and its apply pass clears what the previous call left in the other (stream-ordered, single stream: like _stats_scratch).
    ``dirty[i]``: doubles of buffer i that hold sums.

    hipGraph captures make the host-side bookkeeping blind (a replay runs its calls without this object seeing them), so:
    the FIRST call of every capture clears both buffers inside the capture (one memset pair per replay: the replayed
    sequence is self-contained), and once any capture exists every EAGER call clears both buffers first (correct after any
    interleaving of replays and eager calls; only mixed capture / eager use pays for it).

    Streams: the pair is shared by every stream of the device.  Eager calls from a stream other than the previous
    eager caller's first wait for that stream (``wait_stream``: everything the previous user enqueued, its apply pass
    included, completes before this call's atomics start), so blocks driven from two streams by ONE host thread — a
    side-stream evaluation during training — serialise on the pair instead of mixing their sums.  Concurrent host THREADS
    are not supported (like the rest of the fused path's per-device scratch).  A failed launch between acquire() and the
    apply pass leaves sums behind that the bookkeeping calls clean: callers report it through ``poison()`` and the next
    acquire() clears both buffers."""
    DOUBLES = 2 * 4096 * 4
    _inst = {}

    def __init__(self, device):
        self.store = torch.zeros((2, self.DOUBLES), dtype=torch.float64, device=device)
        self.bufs = [self.store[0], self.store[1]]
        self.cur = 0
        self.dirty = [0, 0]
        self.capture_id = 0
        self.captured = False
        self.poisoned = False
        self.last_stream = None
        self.last_raw = None
        self.hold = 0                   # one_call(): 1 = the next acquire may reset, 2 = later acquires of the same call may not
        self.pending = None             # a _BnLink whose consumer has accumulated into a buffer while the clear of the other one
                                        # waits for the producer's apply pass (fused._dgrad_x6_with_link)

    @classmethod
    def of(cls, device):
        key = (device.type, device.index)
        inst = cls._inst.get(key)
        if inst is None:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.GkgError("BN scratch first used inside a hipGraph capture; run one eager warm-up step first")
            inst = cls._inst[key] = cls(device)
        return inst

    def acquire(self, lib, n):
        """-> (buffer to accumulate into: clean, buffer to clear, doubles to clear); the caller's kernels do the clearing."""
        cap = lib.gkg_stream_capture_id(_stream()) if torch.cuda.is_current_stream_capturing() else 0
        if cap:
            self.captured = True
            if cap != self.capture_id:
                self.capture_id = cap
                self._reset()
        else:
            raw = _stream()
            if self.last_raw != raw:                    # (Stream objects only when the caller's stream actually changed)
                here = torch.cuda.current_stream(self.store.device)
                if self.last_stream is not None and self.last_stream != here:
                    here.wait_stream(self.last_stream)
                self.last_stream = here
                self.last_raw = raw
            if (self.captured or self.poisoned) and self.hold != 2:
                self._reset()
        if self.pending is not None:
            # an acquire between a consumer's dgrad epilogue and its producer's apply pass (another ready BN backward node was
            # scheduled in between, or the producer's backward never ran: autograd.grad stopping at the activation): the
            # deferred clear has not happened and the bookkeeping calls that buffer clean -> start clean, and the
            # producer falls back to its own statistics pass
            self.pending.ready = None
            self.pending = None
            self._reset()
        cur, other = self.bufs[self.cur], self.bufs[self.cur ^ 1]
        zero = self.dirty[self.cur ^ 1]
        self.dirty[self.cur], self.dirty[self.cur ^ 1] = n, 0
        self.cur ^= 1
        if self.hold == 1:
            self.hold = 2
        return cur, other, zero

    @contextlib.contextmanager
    def one_call(self):
        """Several acquire() calls whose launches are all issued afterwards, together (the block driver fills every layer's
        descriptor first): the clear that an eager call owes after a capture or a failed launch is enqueued by the FIRST
        acquire only — a later one would clear, ahead of the block's launches, a buffer the bookkeeping then hands to a layer
        while an earlier layer's sums are still in it."""
        self.hold = 1
        try:
            yield self
        finally:
            self.hold = 0

    def fold_reset(self, cap):
        """The first use of the pair inside hipGraph capture ``cap`` would clear both buffers with a launch of its own: when the
        weight-plane refresh (the first launch of a captured step) runs first, it clears them in ITS launch — returns the storage
        to clear, or None when this capture has already been seen."""
        if not cap or cap == self.capture_id:
            return None
        self.captured = True
        self.capture_id = cap
        self.dirty = [0, 0]
        self.poisoned = False
        if self.pending is not None:
            self.pending.ready = None
            self.pending = None
        return self.store

    def _reset(self):
        self.store.fill_(0.0)            # ONE elementwise launch (a captured memset node measured far slower than a kernel node)
        self.dirty = [0, 0]
        self.poisoned = False
        if self.pending is not None:
            self.pending.ready = None
            self.pending = None

    def poison(self):
        """A launch between acquire() and its apply pass failed: the buffers' contents are unknown."""
        self.poisoned = True


_BnFwdScratch = _BnBwdScratch       # forward and backward calls alternate through the SAME pair (one reset per capture)
