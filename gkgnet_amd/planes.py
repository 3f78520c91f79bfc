"""bf16 operand planes of the projection weights for the split-bf16 (x6) GEMM kernels: one registry per device, refreshed when
a parameter's version counter moves, or once per hipGraph capture (csrc/gkg_gemm_x6.hip: x6_prep_kernel)."""
from __future__ import annotations

import ctypes
import weakref

import torch

from . import _lib
from .ops import _ptr, _stream


# Parameter updates that move no version counter — torch.optim's fused=True kernels (measured: AdamW(fused=True).step()
# leaves p._version where it was), updates through p.data, third-party fused optimisers — would leave every cache that is
# keyed on ._version stale (weight planes, bf16 / BN-folded weight copies).  A process-wide optimiser-step counter is part
# of those keys: every torch.optim step bumps it through the global post-step hook; anything else that rewrites
# parameters behind autograd's back calls mark_parameters_updated().
_STEP = [0]


def _on_optimizer_step(*_args, **_kwargs):
    _STEP[0] += 1


from torch.optim.optimizer import register_optimizer_step_post_hook as _register_step_hook      # noqa: E402

_register_step_hook(_on_optimizer_step)


def mark_parameters_updated():
    """Tell the library that parameters were modified in a way autograd's version counters did not see."""
    _STEP[0] += 1


def param_version(t):
    """Staleness token of a parameter-derived cache: (version counter, optimiser-step counter)."""
    return (t._version, _STEP[0])


class _WeightPlanes:
    """bf16 hi / mid / lo planes (forward and dgrad orientation) of every projection weight that went through the x6
    kernels on one device.  The planes are a function of the parameter values only, so they are refreshed when a
    parameter's version counter moves (an optimiser step) — ALL registered weights in ONE launch (gkg_x6_prep_weights).
    Inside a hipGraph capture the host cannot see later in-place updates, so the first projection of each capture emits
    the refresh unconditionally: a captured training step re-splits the weights once per replay.  Once ANY capture has
    gone through this registry the version counters prove nothing on the eager side either — a replay (with an in-graph
    optimiser step) moves the weights after its own re-split and bumps no counter — so from then on (``captured`` is
    sticky) every eager projection re-splits ITS OWN weight right before use (a one-descriptor launch)."""

    def __init__(self, device):
        self.device = device
        self.entries = {}            # id(weight) -> dict
        self.descs = None            # device copy of the descriptor table
        self.solo = None             # the same descriptors, each numbered from unit 0 (single-weight launches)
        self.unit_ends = []
        self.capture_id = 0
        self.captured = False
        self.pending_zero = None     # a buffer (the flat gradient arena) to clear in the next refresh launch of this capture

    def defer_zero(self, t) -> bool:
        """Inside a hipGraph capture whose weight-plane refresh is still to come (it is the first launch of a captured step that
        uses an x6 projection): take over the clearing of ``t`` — it rides in the refresh launch.  False: the caller clears it."""
        if not (ZERO_FOLD and torch.cuda.is_current_stream_capturing() and self.entries and self.pending_zero is None
                and t.is_contiguous() and t.data_ptr() % 16 == 0 and (t.numel() * t.element_size()) % 16 == 0):
            return False
        cap = _lib.load().gkg_stream_capture_id(_stream())
        if not cap or cap == self.capture_id:
            return False
        self.pending_zero = (cap, t)
        return True

    def flush_zero(self):
        """A deferred clear that no refresh has picked up (no x6 projection ran in this capture before a gradient was needed)."""
        if self.pending_zero is not None:
            _, t = self.pending_zero
            self.pending_zero = None
            t.zero_()

    def _register(self, lib, weight, nb, cout, cin, need_f, need_d, old=None, kperm=0):
        if torch.cuda.is_current_stream_capturing():
            raise _lib.GkgError("x6 projection: a weight was first seen inside a hipGraph capture; run one eager "
                                "warm-up step before capturing")
        same = (old is not None and old["ref"]() is weight and old["ptr"] == weight.data_ptr()
                and (old["nb"], old["cout"], old["cin"], old["kperm"]) == (nb, cout, cin, kperm))
        e = dict(ref=weakref.ref(weight), nb=nb, cout=cout, cin=cin, kperm=kperm, ptr=weight.data_ptr(), version=-1,
                 pf=old["pf"] if same else None, pd=old["pd"] if same else None)
        if need_f and e["pf"] is None:
            e["pf"] = torch.empty(lib.gkg_x6_planes_bytes(cin, cout, nb, 0), dtype=torch.uint8, device=self.device)
        if need_d and e["pd"] is None:
            e["pd"] = torch.empty(lib.gkg_x6_planes_bytes(cin, cout, nb, 1), dtype=torch.uint8, device=self.device)
        self.entries[id(weight)] = e
        self.descs = None
        return e

    def _build_descs(self, lib):
        live = {k: e for k, e in self.entries.items() if e["ref"]() is not None}
        self.entries = live
        size = lib.gkg_x6_prep_desc_bytes()
        host = ctypes.create_string_buffer(size * max(1, len(live)))
        units = 0
        self.unit_ends = []
        for i, e in enumerate(live.values()):
            if i % 256 == 0:
                units = 0                                             # unit numbering restarts with every launch's table
            units = lib.gkg_x6_prep_desc_fill(host, i, e["ptr"], _ptr(e["pf"]), _ptr(e["pd"]), e["cin"], e["cout"], e["nb"],
                                              units, e["kperm"])
            if units < 0:
                raise _lib.GkgError("gkg_x6_prep_desc_fill rejected a weight")
            self.unit_ends.append(units)
        self.descs = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(self.device)
        solo = ctypes.create_string_buffer(size * max(1, len(live)))
        for i, e in enumerate(live.values()):
            e["slot"] = i
            e["solo_units"] = lib.gkg_x6_prep_desc_fill(solo, i, e["ptr"], _ptr(e["pf"]), _ptr(e["pd"]), e["cin"], e["cout"],
                                                        e["nb"], 0, e["kperm"])
        self.solo = torch.frombuffer(bytearray(solo.raw), dtype=torch.uint8).to(self.device)

    def refresh_one(self, lib, e):
        """Re-split one registered weight (eager use after a capture: see the class docstring)."""
        if self.descs is None:
            self._build_descs(lib)
        size = lib.gkg_x6_prep_desc_bytes()
        _lib.check(lib.gkg_x6_prep_weights(self.solo.data_ptr() + e["slot"] * size, 1, e["solo_units"], _stream()),
                   "gkg_x6_prep_weights")

    def refresh(self, lib, zero=()):
        """Re-split every registered weight (one launch); ``zero``: up to two tensors cleared by the same launch."""
        if self.descs is None:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.GkgError("x6 weight planes: descriptor table is stale inside a capture; run a warm-up step first")
            self._build_descs(lib)
        if not self.entries:
            return
        n = len(self.entries)
        size = lib.gkg_x6_prep_desc_bytes()
        zero = [t for t in zero if t is not None]
        for i0 in range(0, n, 256):                                   # at most 256 descriptors per launch
            i1 = min(n, i0 + 256)
            if zero and i0 == 0:
                z = [(t.data_ptr(), t.numel() * t.element_size()) for t in zero[:2]] + [(None, 0)] * (2 - len(zero[:2]))
                _lib.check(lib.gkg_x6_prep_weights_zero(self.descs.data_ptr(), i1, self.unit_ends[i1 - 1], z[0][0], z[0][1], z[1][0],
                                                        z[1][1], _stream()), "gkg_x6_prep_weights_zero")
                continue
            _lib.check(lib.gkg_x6_prep_weights(self.descs.data_ptr() + i0 * size, i1 - i0, self.unit_ends[i1 - 1], _stream()),
                       "gkg_x6_prep_weights")
        for e in self.entries.values():
            w = e["ref"]()
            e["version"] = -1 if w is None else param_version(w)

    def get(self, lib, weight, nb, cout, cin, need_f=True, need_d=True, kperm=0):
        """``kperm``: the grouped projection behind the aggregation — input columns as [x chunk | m chunk] (gkg_hip.h "XM layout")."""
        e = self.entries.get(id(weight))
        if (e is None or e["ref"]() is not weight or e["ptr"] != weight.data_ptr()
                or (e["nb"], e["cout"], e["cin"], e["kperm"]) != (nb, cout, cin, kperm) or (need_f and e["pf"] is None)
                or (need_d and e["pd"] is None)):
            e = self._register(lib, weight, nb, cout, cin, need_f, need_d, e, kperm)
        cap = lib.gkg_stream_capture_id(_stream()) if torch.cuda.is_current_stream_capturing() else 0
        if cap:
            self.captured = True
            if cap != self.capture_id:
                self.capture_id = cap
                # the first launch of a captured step: it also clears what the step clears before its first projection anyway
                # — the gradient arena handed over by GradBucket.release(prezero=True) and the fp64 BN scratch pair
                zero = []
                if self.pending_zero is not None:
                    pcap, t = self.pending_zero
                    self.pending_zero = None
                    if pcap == cap:
                        zero.append(t)
                    else:
                        t.zero_()
                if ZERO_FOLD:
                    from .bn_scratch import _BnBwdScratch
                    scr = _BnBwdScratch._inst.get((self.device.type, self.device.index))
                    if scr is not None:
                        zero.append(scr.fold_reset(cap))
                self.refresh(lib, zero)
        elif e["version"] != param_version(weight):
            self.refresh(lib)                    # eager call, stale by the version / step counters (an eager optimiser step)
        elif self.captured:
            # graphs exist: any replay since the last eager call may have moved this weight (a captured optimiser step)
            # without a counter moving, and it may do so again between any two eager calls
            self.refresh_one(lib, e)
        return e["pf"], e["pd"]


_PLANES = {}
ZERO_FOLD = True       # the gradient arena and the BN scratch are cleared by the weight-plane refresh launch of a captured step


def defer_zero(t) -> bool:
    """GradBucket.release(prezero=True): let the weight-plane refresh of the captured step clear ``t`` (see _WeightPlanes)."""
    reg = _PLANES.get((t.device.type, t.device.index))
    return reg is not None and reg.defer_zero(t)


def flush_deferred_zero():
    for reg in _PLANES.values():
        reg.flush_zero()


def _planes(lib, weight, nb, cout, cin, need_f=True, need_d=True, kperm=0):
    key = (weight.device.type, weight.device.index)
    reg = _PLANES.get(key)
    if reg is None:
        reg = _PLANES[key] = _WeightPlanes(weight.device)
    return reg.get(lib, weight, nb, cout, cin, need_f, need_d, kperm)


def refresh_weight_planes(device=None):
    """Re-split the registered projection weights now (normally automatic: see _WeightPlanes)."""
    lib = _lib.load()
    for key, reg in _PLANES.items():
        if device is None or key == (torch.device(device).type, torch.device(device).index):
            reg.refresh(lib)
