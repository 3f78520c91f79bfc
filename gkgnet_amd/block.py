"""Block-level host side (round 6): a Grapher / GrapherLabel block's forward and backward as ONE library call each
(csrc/gkg_block.hip: gkg_grapher_fwd / _bwd, gkg_grapher_label_fwd / _bwd; reference torch_vertex.py:325-333, :392-403).

``fused.py`` composes a block from per-layer autograd Functions (≈ 25 ctypes calls and ≈ 1 800 Python calls per block pair and
step): right for the whole-backbone steps, which are GPU-bound, and host-bound for the small blocks an eager training loop launches
one by one.  Here the host allocates two arenas (what the forward saves, what the backward needs), fills a descriptor of pointers
and sizes and makes one call; the C side issues the same launches with the same arguments, so the results are bit-identical to the
composition (tests/test_hip_block_driver.py).  Taken automatically for the form the metric is quoted on — fp32, train-mode
BatchNorm with rank-local statistics, no DropPath scaling, un-pooled keys, every projection on the split-bf16 kernels, blocks
below the BN-epilogue row count; everything else keeps the composition.  GKG_DISABLE=block_driver: off."""
from __future__ import annotations

import ctypes as C
import weakref

import torch

from . import _lib, fused
from .ops import _ptr, _stream

_F32 = torch.float32
V, I, Z, F, U = C.c_void_p, C.c_int, C.c_size_t, C.c_float, C.c_uint


class ProjBN(C.Structure):
    _fields_ = [("planes_fwd", V), ("planes_dgrad", V), ("gamma", V), ("beta", V), ("bias", V), ("running_mean", V), ("running_var", V),
                ("nbt", V), ("momentum", F), ("eps", F), ("cin", I), ("cout", I), ("nb", I), ("fsum", V), ("fzero", V), ("fzero_n", Z),
                ("bsum", V), ("bzero", V), ("bzero_n", Z), ("Y", V), ("bn", V), ("dw", V), ("dgamma", V), ("dbeta", V)]


class GraphOp(C.Structure):
    _fields_ = [("G", I), ("k", I), ("d", I), ("fused_mr", I), ("relpos", V), ("knn_flags", U), ("mr_flags", U), ("knn_ws", V),
                ("knn_ws_bytes", Z), ("arg", V), ("nn16", V), ("nn_idx", V), ("center", V)]


class GrapherBlock(C.Structure):
    _fields_ = [("B", I), ("C", I), ("H", I), ("W", I), ("x", V), ("out", V), ("out_tm", V), ("xt", V), ("XM", V), ("A2", V),
                ("fc1", ProjBN), ("conv", ProjBN), ("fc2", ProjBN), ("graph", GraphOp), ("sk_ws", V), ("sk_bytes", Z),
                ("keys_G", I), ("keys_L", I), ("keys_k", I), ("keys_d", I), ("keys_fused_mr", I), ("keys_flags", U), ("keys_ws", V),
                ("keys_ws_bytes", Z), ("dout", V), ("dout_tm", V), ("dx", V), ("g3", V), ("dY3", V), ("dA2", V), ("dY2", V), ("dXM", V),
                ("gx1", V), ("dY1", V), ("dxt", V)]


class LabelBlock(C.Structure):
    _fields_ = [("B", I), ("C", I), ("L", I), ("M", I), ("e", V), ("ft", V), ("out", V), ("XM", V), ("A2", V), ("h2", V), ("f1", V),
                ("fc1", ProjBN), ("conv", ProjBN), ("fc2", ProjBN), ("ffn1", ProjBN), ("ffn2", ProjBN), ("graph", GraphOp), ("sk_ws", V),
                ("sk_bytes", Z), ("dout", V), ("de", V), ("dft", V), ("dY5", V), ("df1", V), ("dY4", V), ("dh2", V), ("dY3", V),
                ("dA2", V), ("dY2", V), ("dXM", V), ("gx1", V), ("dY1", V)]


ENABLED = "block_driver" not in fused._DISABLED
_BOUND = False


def _bind(lib):
    global _BOUND
    if _BOUND:
        return
    lib.gkg_grapher_fwd.restype = I
    lib.gkg_grapher_fwd.argtypes = [C.POINTER(GrapherBlock), V]
    lib.gkg_grapher_bwd.restype = I
    lib.gkg_grapher_bwd.argtypes = [C.POINTER(GrapherBlock), C.POINTER(_lib.WgradProblem), V]
    lib.gkg_grapher_label_fwd.restype = I
    lib.gkg_grapher_label_fwd.argtypes = [C.POINTER(LabelBlock), V]
    lib.gkg_grapher_label_bwd.restype = I
    lib.gkg_grapher_label_bwd.argtypes = [C.POINTER(LabelBlock), C.POINTER(_lib.WgradProblem), V]
    _BOUND = True


# ----------------------------------------------------------------------------------------------- eligibility
def _proj_ok(seq, R, cin, cout, nb) -> bool:
    conv, bn = seq[0], seq[1]
    return (conv.weight.dtype == _F32 and bn.weight.dtype == _F32 and bn.training and fused._bn_ok(bn) and fused._sync_group(bn) is None
            and isinstance(bn, torch.nn.modules.batchnorm._BatchNorm)
            and fused._x6_rule(R, cin, cout, nb, "fwd") and fused._derive_ok(bn, nb, cout, _lib.F32, False)
            and (conv.bias is None or conv.bias.dtype == _F32))


def _common_ok(x) -> bool:
    return (ENABLED and fused.ENABLED and fused.GEMM_MATH == "x6" and not fused.DETERMINISTIC and fused.XM_DIRECT
            and x.is_cuda and x.dtype == _F32 and not torch.is_autocast_enabled() and torch.is_grad_enabled()
            and fused.knn_graph_tm is fused._KNN_GRAPH_TM)


def _drops(dp) -> bool:
    """An ACTIVE DropPath (asked without drawing: the eligibility test must not consume random numbers)."""
    if hasattr(dp, "active"):
        return bool(dp.active())
    return not isinstance(dp, torch.nn.Identity)


def grapher_ok(mod, x, relative_pos, groups, want_edge, dual) -> bool:
    """The block driver applies to this Grapher call (see the module docstring); ``fused.grapher_forward`` asks."""
    if not _common_ok(x) or want_edge or x.dim() != 4 or fused.is_channels_last(x):
        return False
    gc = mod.graph_conv
    B, Cc, H, W = x.shape
    T = B * H * W
    nn_ = gc.gconv.nn
    if (gc.r != 1 or Cc % 16 or (Cc // groups) % 4 or T >= fused.BN_EPILOGUE_MIN_ROWS or len(nn_) != 3
            or _drops(mod.drop_path) or H * W > 65536):
        return False
    if not (_proj_ok(mod.fc1, T, Cc, Cc, 1) and _proj_ok(nn_, T, Cc // 2, Cc // 2, 4) and _proj_ok(mod.fc2, T, 2 * Cc, Cc, 1)):
        return False
    if nn_[0].groups != 4 or tuple(nn_[0].weight.shape[:2]) != (2 * Cc, Cc // 2) or not isinstance(nn_[2], torch.nn.GELU):
        return False
    fm = fused._knn_mr_shapes_ok(B, H * W, Cc, H * W, False, relative_pos, gc.k, gc.d, groups, nn_, False)
    return bool(fm or fused.KNN_COMPACT)


def label_ok(mod, e, ft, groups) -> bool:
    if not _common_ok(e) or ft.dtype != _F32 or not ft.is_contiguous():
        return False
    gc = mod.graph_conv
    B, L, Cc = e.shape
    T = B * L
    M = ft.shape[1]
    nn_ = gc.gconv.nn
    Cf = mod.ffn.fc1[0].weight.shape[0]
    if (Cc % 16 or (Cc // groups) % 4 or T >= fused.BN_EPILOGUE_MIN_ROWS or len(nn_) != 3 or M > 65536
            or _drops(mod.drop_path) or _drops(mod.ffn.drop_path)
            or not isinstance(mod.ffn.act, torch.nn.GELU) or not isinstance(nn_[2], torch.nn.GELU)):
        return False
    if nn_[0].groups != 4 or tuple(nn_[0].weight.shape[:2]) != (2 * Cc, Cc // 2):
        return False
    return (_proj_ok(mod.fc1, T, Cc, Cc, 1) and _proj_ok(nn_, T, Cc // 2, Cc // 2, 4) and _proj_ok(mod.fc2, T, 2 * Cc, Cc, 1)
            and _proj_ok(mod.ffn.fc1, T, Cc, Cf, 1) and _proj_ok(mod.ffn.fc2, T, Cf, Cc, 1))


# ----------------------------------------------------------------------------------------------- plans
# What a block call needs to know about its module is the same on every step: which tensors the five projections read, their
# sizes, the layout of the two arenas, the static half of the descriptor.  A _Plan holds it (built on the first eligible call,
# through the full eligibility test above) and a per-call guard that is a few dozen identity / pointer comparisons; Grapher.forward
# and GrapherLabel.forward ask try_grapher / try_label first and reach the C entry point after ~40 us of Python instead of ~150.
_PLANS = weakref.WeakKeyDictionary()           # module -> {(shape key, switches): _Plan}


def _switches():
    return (ENABLED, fused.ENABLED, fused.GEMM_MATH, fused.DETERMINISTIC, fused.XM_DIRECT, fused.KNN_MR, fused.KNN_COMPACT,
            fused.KNN_PREP, fused.KNN_BF16, fused.BN_EPILOGUE_MIN_ROWS, fused.knn_graph_tm is fused._KNN_GRAPH_TM)


class _Proj:
    """One 1x1 projection + BN of a block: the tensors its kernels read through raw pointers, and its sizes."""
    __slots__ = ("conv", "bn", "W", "bias", "gamma", "beta", "rm", "rv", "nbt", "nb", "cin", "cout", "kperm", "trs", "mom", "eps",
                 "wshape", "nch", "wreal")

    def __init__(self, seq, nb, cin, cout, kperm, wshape, ident):
        conv, bn = seq[0], seq[1]
        self.conv, self.bn = conv, bn
        self.W, self.bias, self.gamma, self.beta = conv.weight, conv.bias, bn.weight, bn.bias
        self.trs = bool(bn.track_running_stats)
        self.rm, self.rv, self.nbt = bn.running_mean, bn.running_var, bn.num_batches_tracked
        self.nb, self.cin, self.cout, self.kperm = nb, cin, cout, kperm
        self.mom, self.eps = bn.momentum, bn.eps
        self.wshape, self.nch, self.wreal = wshape, nb * cout, conv.weight.shape
        ident += [(seq._modules, "0", conv), (seq._modules, "1", bn), (conv._parameters, "weight", self.W),
                  (conv._parameters, "bias", self.bias), (bn._parameters, "weight", self.gamma), (bn._parameters, "bias", self.beta),
                  (bn._buffers, "running_mean", self.rm), (bn._buffers, "running_var", self.rv),
                  (bn._buffers, "num_batches_tracked", self.nbt)]

    def static(self, p: ProjBN):
        track = self.trs
        p.gamma, p.beta, p.bias = _ptr(self.gamma), _ptr(self.beta), _ptr(self.bias)
        p.running_mean = _ptr(self.rm) if track else None
        p.running_var = _ptr(self.rv) if track else None
        p.nbt = _ptr(self.nbt) if track else None
        p.momentum, p.eps = float(self.mom), float(self.eps)
        p.cin, p.cout, p.nb = self.cin, self.cout, self.nb

    def baked(self):
        return [t for t in (self.gamma, self.beta, self.bias) + ((self.rm, self.rv, self.nbt) if self.trs else ()) if t is not None]


def _layout(sizes):
    """Arena layout: element offsets of 16-byte aligned fp32 pieces -> (offsets, total elements)."""
    offs, o = [], 0
    for n in sizes:
        offs.append(o)
        o += (n + 3) & ~3
    return offs, o


class _Plan:
    __slots__ = ("kind", "projs", "ident", "tensors", "ptrs", "tmpl", "fwd_offs", "fwd_total", "bwd_offs", "bwd_total", "drops",
                 "sync", "gc", "nn_", "k", "d", "groups", "dims", "rp", "rp_view", "fast", "params", "fm", "has_bucket", "__weakref__")

    def valid(self) -> bool:
        for dct, key, obj in self.ident:
            if dct.get(key) is not obj:
                return False
        for p in self.projs:
            bn = p.bn
            if (not bn.training or bn.track_running_stats != p.trs or bn.momentum != p.mom or bn.eps != p.eps
                    or p.W.shape != p.wreal or p.W.dtype != _F32):
                return False
        if self.ptrs != [t.data_ptr() for t in self.tensors]:
            return False
        for dp in self.drops:
            if _drops(dp):
                return False
        gc = self.gc
        if gc.k != self.k or gc.d != self.d:
            return False
        if self.sync:
            for p in self.projs:
                if fused._sync_group(p.bn) is not None:
                    return False
        return True


def _finish_plan(plan, mod, cls, projs, ident, drops, gc, nn_, groups, dims, relative_pos, fwd_sizes, bwd_sizes):
    plan.projs, plan.ident, plan.drops, plan.gc, plan.nn_, plan.groups, plan.dims = projs, ident, drops, gc, nn_, groups, dims
    plan.k, plan.d = gc.k, gc.d
    plan.tensors = [t for p in projs for t in p.baked()]
    plan.ptrs = [t.data_ptr() for t in plan.tensors]
    plan.sync = any(isinstance(p.bn, torch.nn.SyncBatchNorm) for p in projs)
    plan.fwd_offs, plan.fwd_total = _layout(fwd_sizes)
    plan.bwd_offs, plan.bwd_total = _layout(bwd_sizes)
    plan.params = tuple(t for p in projs for t in (p.W, p.gamma, p.beta))
    plan.fm = {}
    own = mod._parameters.get("relative_pos", mod.__dict__.get("relative_pos"))
    plan.rp = relative_pos
    plan.fast = relative_pos is None or relative_pos is own        # a re-interpolated bias is a new tensor every call: slow path
    plan.rp_view = None
    if relative_pos is not None and plan.fast:
        ident.append((mod._parameters, "relative_pos", relative_pos))
    d = cls()
    names = [f[0] for f in cls._fields_ if f[1] is ProjBN]
    for nm, p in zip(names, projs):
        p.static(getattr(d, nm))
    plan.tmpl = bytes(d)
    return plan


def _plan_grapher(mod, x, relative_pos, groups):
    plans = _PLANS.setdefault(mod, {})
    key = ("g", tuple(x.shape), groups, _switches())
    plan = plans.get(key)
    if plan is not None and plan.valid() and (plan.rp is relative_pos or not plan.fast):
        return plan
    B, Cc, H, W = x.shape
    T = B * H * W
    gc = mod.graph_conv
    nn_ = gc.gconv.nn
    ident = [(mod._modules, "fc1", mod.fc1), (mod._modules, "fc2", mod.fc2), (mod._modules, "graph_conv", gc),
             (mod._modules, "drop_path", mod.drop_path), (gc._modules, "gconv", gc.gconv), (gc.gconv._modules, "nn", nn_)]
    projs = [_Proj(mod.fc1, 1, Cc, Cc, 0, (Cc, Cc), ident), _Proj(nn_, 4, Cc // 2, Cc // 2, 1, (4, Cc // 2, Cc // 2), ident),
             _Proj(mod.fc2, 1, 2 * Cc, Cc, 0, (Cc, 2 * Cc), ident)]
    plan = _Plan()
    plan.kind = "g"
    # forward arena: xt, XM, A2, Y1, Y2, Y3, bn1, bn2, bn3, winning rows (u16)   backward: g3, dY3, gx1, dY1, dxt, dA2, dY2, dXM
    fwd = [T * Cc, T * 2 * Cc, T * 2 * Cc, T * Cc, 4 * T * (Cc // 2), T * Cc, 4 * Cc, 8 * Cc, 4 * Cc, T * (Cc // 2)]
    bwd = [T * Cc] * 5 + [T * 2 * Cc] * 3
    _finish_plan(plan, mod, GrapherBlock, projs, ident, [mod.drop_path], gc, nn_, groups, (B, Cc, H, W), relative_pos, fwd, bwd)
    plans[key] = plan
    return plan


def _plan_label(mod, e2, ft, groups):
    plans = _PLANS.setdefault(mod, {})
    B, M, Cc = ft.shape
    T = e2.shape[0]
    key = ("l", T, tuple(ft.shape), groups, _switches())
    plan = plans.get(key)
    if plan is not None and plan.valid():
        return plan
    L = T // B
    gc = mod.graph_conv
    nn_ = gc.gconv.nn
    ffn = mod.ffn
    Cf = ffn.fc1[0].weight.shape[0]
    ident = [(mod._modules, "fc1", mod.fc1), (mod._modules, "fc2", mod.fc2), (mod._modules, "graph_conv", gc),
             (mod._modules, "drop_path", mod.drop_path), (mod._modules, "ffn", ffn), (ffn._modules, "fc1", ffn.fc1),
             (ffn._modules, "fc2", ffn.fc2), (ffn._modules, "drop_path", ffn.drop_path), (ffn._modules, "act", ffn.act),
             (gc._modules, "gconv", gc.gconv), (gc.gconv._modules, "nn", nn_)]
    projs = [_Proj(mod.fc1, 1, Cc, Cc, 0, (Cc, Cc), ident), _Proj(nn_, 4, Cc // 2, Cc // 2, 1, (4, Cc // 2, Cc // 2), ident),
             _Proj(mod.fc2, 1, 2 * Cc, Cc, 0, (Cc, 2 * Cc), ident), _Proj(ffn.fc1, 1, Cc, Cf, 0, (Cf, Cc), ident),
             _Proj(ffn.fc2, 1, Cf, Cc, 0, (Cc, Cf), ident)]
    plan = _Plan()
    plan.kind = "l"
    # forward arena: XM, A2, h2, f1, Y1..Y5, bn1..bn5, winning rows      backward: dY5, dh2, dY3, gx1, dY1, df1, dY4, dA2, dY2, dXM
    fwd = [T * 2 * Cc, T * 2 * Cc, T * Cc, T * Cf, T * Cc, 4 * T * (Cc // 2), T * Cc, T * Cf, T * Cc, 4 * Cc, 8 * Cc, 4 * Cc, 4 * Cf,
           4 * Cc, T * (Cc // 2)]
    bwd = [T * Cc] * 5 + [T * Cf] * 2 + [T * 2 * Cc] * 3
    _finish_plan(plan, mod, LabelBlock, projs, ident, [mod.drop_path, ffn.drop_path], gc, nn_, groups, (B, Cc, L, M, Cf), None, fwd, bwd)
    plans[key] = plan
    return plan


# ----------------------------------------------------------------------------------------------- descriptor pieces
def _proj_fwd(lib, p: ProjBN, pr: _Proj, scratch, keep, y_ptr, bn_ptr):
    """The per-call half of a projection's forward descriptor: weight planes (refreshed when the weight moved), the BN pass's
    scratch buffers, where Y and the BN coefficients go."""
    pf, pd = fused._planes(lib, pr.W, pr.nb, pr.cout, pr.cin, True, True, kperm=pr.kperm)
    if pr.trs:
        ep = pr.bn.__dict__.get("_gkg_epoch")
        if ep is not None:
            pr.bn.__dict__["_gkg_epoch"] = ep + 1             # fused._touch_stats: the kernels update the running statistics
    cur, other, zero = scratch.acquire(lib, 2 * pr.nch)
    p.planes_fwd, p.planes_dgrad = pf.data_ptr(), pd.data_ptr()
    p.fsum, p.fzero, p.fzero_n = cur.data_ptr(), other.data_ptr(), zero
    p.Y, p.bn = y_ptr, bn_ptr
    keep.append(pf)
    keep.append(pd)


def _proj_bwd(lib, p: ProjBN, pr: _Proj, scratch, dev):
    """Backward half: the BN pass's scratch buffers and the gradient outputs (bucket slots when the parameters have them)."""
    dWv, dgamma, dbeta = fused._grad_outs((pr.W, pr.gamma, pr.beta), pr.wshape, pr.nch, dev)
    if not getattr(dWv, "_gkg_zero", False):
        dWv.zero_()                                      # the weight-gradient kernels ADD into dw
    cur, other, zero = scratch.acquire(lib, 2 * pr.nch)
    p.bsum, p.bzero, p.bzero_n = cur.data_ptr(), other.data_ptr(), zero
    p.dw, p.dgamma, p.dbeta = dWv.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr()
    return dWv, dgamma, dbeta


def _graph_op(lib, plan, g: GraphOp, B, G, c, N, M, relative_pos, has_y, want_edge, dev, keys_key, keep):
    """The block's k-NN + aggregation: kernel form, flags, workspace (shared with a keys producer when the Grapher in front prepared
    this graph's keys).  -> (key object of this k-NN problem, edge tensor | None)."""
    k, d = plan.k, plan.d
    flags0 = _lib.KNN_NORMALIZE | _lib.knn_select_flags() | _lib.relpos_flags(relative_pos)
    fm = plan.fm.get(flags0)
    if fm is None:
        fm = plan.fm[flags0] = bool(fused._knn_mr_shapes_ok(B, N, G * c, M, has_y, relative_pos, k, d, G, plan.nn_, False))
    key = fused._KnnKey(B, G, c, N, M, k, d, has_y, relative_pos, fm, flags0)
    flags = flags0
    if fused.KNN_PREP:
        flags |= _lib.KNN_X_PREPARED
        if keys_key is not None and keys_key.ws is not None and keys_key.tuple() == key.tuple():
            key.ws, key.y_ready = keys_key.ws, True
            flags |= _lib.KNN_Y_PREPARED
    if key.ws is None:
        key.ws = fused._ws(lib.gkg_knn_workspace_bytes(B * G, c, N, M, k, d, _lib.F32, _lib.KNN_NORMALIZE), dev)
    rp = None
    if relative_pos is not None:
        rp = plan.rp_view if relative_pos is plan.rp else None
        if rp is None or rp.data_ptr() != relative_pos.data_ptr():
            rp = fused._rp_arg(relative_pos, N, M)
            if relative_pos is plan.rp and plan.fast and rp.data_ptr() == relative_pos.data_ptr():
                plan.rp_view = rp                      # a reshaped view of the module's own parameter: the same view every call
    g.G, g.k, g.d, g.fused_mr = G, k, d, int(fm)
    g.relpos, g.knn_flags, g.mr_flags = _ptr(rp), flags, fused._mr_bwd_flags()
    g.knn_ws, g.knn_ws_bytes = key.ws.data_ptr(), key.ws.numel()
    edge = None
    if want_edge:
        edge = torch.empty((2, B * G, N, k), dtype=torch.int64, device=dev)
        g.nn_idx, g.center = edge[0].data_ptr(), edge[1].data_ptr()
    elif not fm:
        nn16 = torch.empty((B * G, N, k), dtype=torch.int16, device=dev)
        g.nn16 = nn16.data_ptr()
        keep.append(nn16)
    keep.append(rp)
    keep.append(key.ws)
    return key, edge


def _issue_wgrads(lib, wq, n, outs, keep, device):
    """The block's weight-gradient problems: into the backward pass's batched launch when every dW is a bucket slot (and a backward
    pass is running to flush it), else launched now."""
    if fused._wgrad_defer_block(wq, n, outs, keep, device):
        return
    _lib.check(lib.gkg_linear_wgrad_x6_batch(wq, n, fused.WGRAD_UNITS, _stream()), "gkg_linear_wgrad_x6_batch (block)")


# ----------------------------------------------------------------------------------------------- Grapher
class _GrapherBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, g1, b1, wc, gc_, bc, w2, g2, b2, plan, relative_pos, label_knn, dual):
        lib = _lib.load()
        _bind(lib)
        B, Cc, H, W = plan.dims
        N, T, dev = H * W, B * H * W, x.device
        groups = plan.groups
        x = x.contiguous()
        d = GrapherBlock.from_buffer_copy(plan.tmpl)
        keep = []
        scratch = fused._BnFwdScratch.of(dev)
        buf = torch.empty(plan.fwd_total, dtype=_F32, device=dev)
        base = buf.data_ptr()
        oxt, oXM, oA2, oY1, oY2, oY3, ob1, ob2, ob3, oarg = [base + 4 * o for o in plan.fwd_offs]
        out = torch.empty((B, Cc, H, W), dtype=_F32, device=dev)
        out_tm = torch.empty((T, Cc), dtype=_F32, device=dev) if dual else None
        d.B, d.C, d.H, d.W = B, Cc, H, W
        d.x, d.out, d.out_tm = x.data_ptr(), out.data_ptr(), _ptr(out_tm)
        d.xt, d.XM, d.A2 = oxt, oXM, oA2
        p1, pc, p2 = plan.projs
        scratch.hold = 1                 # every layer's buffers are handed out before the first launch (bn_scratch.one_call)
        try:
            _proj_fwd(lib, d.fc1, p1, scratch, keep, oY1, ob1)
            _proj_fwd(lib, d.conv, pc, scratch, keep, oY2, ob2)
            _proj_fwd(lib, d.fc2, p2, scratch, keep, oY3, ob3)
            _graph_op(lib, plan, d.graph, B, groups, Cc // groups, N, N, relative_pos, False, False, dev, None, keep)
            d.graph.arg = oarg
            sk = fused._sk_ws(dev)
            d.sk_ws, d.sk_bytes = sk.data_ptr(), sk.numel()
            kk = None
            if label_knn is not None and dual and fused.KNN_PREP:
                G2, L2, k2, d2, fm2 = label_knn
                if Cc % G2 == 0 and (Cc // G2) % 4 == 0:
                    kk = fused._KnnKey(B, G2, Cc // G2, L2, N, k2, d2, True, None, fm2)
                    kk.as_keys = 1
                    kk.ws = fused._ws(lib.gkg_knn_workspace_bytes(B * G2, Cc // G2, L2, N, k2, d2, _lib.F32, _lib.KNN_NORMALIZE), dev)
                    d.keys_G, d.keys_L, d.keys_k, d.keys_d, d.keys_fused_mr, d.keys_flags = G2, L2, k2, d2, int(fm2), kk.flags
                    d.keys_ws, d.keys_ws_bytes = kk.ws.data_ptr(), kk.ws.numel()
            _lib.check(lib.gkg_grapher_fwd(C.byref(d), _stream()), "gkg_grapher_fwd")
        except Exception:
            scratch.poison()
            raise
        finally:
            scratch.hold = 0
        if kk is not None:
            out_tm._gkg_knn_keys = kk
        ctx.save_for_backward(buf, w1, wc, w2)
        ctx.desc = d
        ctx.plan = plan
        if dual:
            ctx.set_materialize_grads(False)
            return out, out_tm
        return out

    @staticmethod
    def backward(ctx, dout, dtm=None):
        lib = _lib.load()
        plan = ctx.plan
        B, Cc, H, W = plan.dims
        if dout is None and dtm is None:
            return (None,) * 14
        buf, w1, wc, w2 = ctx.saved_tensors
        T, dev = B * H * W, buf.device
        d = ctx.desc
        tbuf = torch.empty(plan.bwd_total, dtype=_F32, device=dev)
        base = tbuf.data_ptr()
        d.g3, d.dY3, d.gx1, d.dY1, d.dxt, d.dA2, d.dY2, d.dXM = [base + 4 * o for o in plan.bwd_offs]
        dx = torch.empty((B, Cc, H, W), dtype=_F32, device=dev)
        if dout is None:                                          # only the token-major companion was used downstream
            dout = torch.zeros((B, Cc, H, W), dtype=_F32, device=dev)
        dout_c = dout.contiguous()
        dtm_c = None if dtm is None else dtm.contiguous()
        d.dout, d.dout_tm, d.dx = dout_c.data_ptr(), _ptr(dtm_c), dx.data_ptr()
        scratch = fused._BnBwdScratch.of(dev)
        wq = (_lib.WgradProblem * 3)()
        p1, pc, p2 = plan.projs
        scratch.hold = 1                 # every layer's buffers are handed out before the first launch (bn_scratch.one_call)
        try:
            o2 = _proj_bwd(lib, d.fc2, p2, scratch, dev)
            oc = _proj_bwd(lib, d.conv, pc, scratch, dev)
            o1 = _proj_bwd(lib, d.fc1, p1, scratch, dev)
            _lib.check(lib.gkg_grapher_bwd(C.byref(d), wq, _stream()), "gkg_grapher_bwd")
        except Exception:
            scratch.poison()
            raise
        finally:
            scratch.hold = 0
        _issue_wgrads(lib, wq, 3, (o2[0], oc[0], o1[0]), (buf, tbuf, dout_c, dtm_c), dev)
        return (dx, o1[0].view_as(w1), o1[1], o1[2], oc[0].view_as(wc), oc[1], oc[2], o2[0].view_as(w2), o2[1], o2[2], None, None, None, None)


def _run_grapher(plan, mod, x, relative_pos, dual):
    res = _GrapherBlockFn.apply(x, *plan.params, plan, relative_pos, mod.__dict__.get("_gkg_label_knn"), dual)
    out = res[0] if dual else res
    if dual:
        out._gkg_tm = (out._version, res[1])
    if fused.DUAL_LAYOUT:
        out._gkg_producer = weakref.ref(mod)
    return out


def grapher_forward(mod, x, relative_pos, groups, dual):
    """After grapher_ok(): the block through the driver (fused.grapher_forward; builds the plan the next calls go through)."""
    return _run_grapher(_plan_grapher(mod, x, relative_pos, groups), mod, x, relative_pos, dual)


def try_grapher(mod, x):
    """Grapher.forward's first question: a step this module has taken before (same shapes, same switches, same tensors behind the
    same names) goes straight to the driver -> the block's output; None: ask the long way (fused.fused_supported ...)."""
    plans = _PLANS.get(mod)
    if plans is None or not (ENABLED and x.is_cuda and x.dtype == _F32 and torch.is_grad_enabled()
                             and not torch.is_autocast_enabled()):
        return None
    plan = plans.get(("g", tuple(x.shape), mod.graph_conv.num_head, _switches()))
    if plan is None or not plan.fast or not x.is_contiguous() or not plan.valid():
        return None
    return _run_grapher(plan, mod, x, plan.rp, fused.DUAL_LAYOUT and mod.__dict__.get("_gkg_want_tm", False))


# ----------------------------------------------------------------------------------------------- GrapherLabel
class _LabelBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e2, ft, w1, g1, b1, wc, gc_, bc, w2, g2, b2, w4, g4, b4, w5, g5, b5, plan, keys_key, producer):
        lib = _lib.load()
        _bind(lib)
        B, Cc, L, M, Cf = plan.dims
        T, dev = B * L, e2.device
        groups = plan.groups
        d = LabelBlock.from_buffer_copy(plan.tmpl)
        keep = []
        scratch = fused._BnFwdScratch.of(dev)
        buf = torch.empty(plan.fwd_total, dtype=_F32, device=dev)
        base = buf.data_ptr()
        oXM, oA2, oh2, of1, oY1, oY2, oY3, oY4, oY5, ob1, ob2, ob3, ob4, ob5, oarg = [base + 4 * o for o in plan.fwd_offs]
        out = torch.empty((T, Cc), dtype=_F32, device=dev)
        d.B, d.C, d.L, d.M = B, Cc, L, M
        d.e, d.ft, d.out = e2.data_ptr(), ft.data_ptr(), out.data_ptr()
        d.XM, d.A2, d.h2, d.f1 = oXM, oA2, oh2, of1
        scratch.hold = 1                 # every layer's buffers are handed out before the first launch (bn_scratch.one_call)
        try:
            for p, pr, oy, ob in zip((d.fc1, d.conv, d.fc2, d.ffn1, d.ffn2), plan.projs, (oY1, oY2, oY3, oY4, oY5), (ob1, ob2, ob3, ob4, ob5)):
                _proj_fwd(lib, p, pr, scratch, keep, oy, ob)
            key, edge = _graph_op(lib, plan, d.graph, B, groups, Cc // groups, L, M, None, True, True, dev, keys_key, keep)
            d.graph.arg = oarg
            sk = fused._sk_ws(dev)
            d.sk_ws, d.sk_bytes = sk.data_ptr(), sk.numel()
            _lib.check(lib.gkg_grapher_label_fwd(C.byref(d), _stream()), "gkg_grapher_label_fwd")
        except Exception:
            scratch.poison()
            raise
        finally:
            scratch.hold = 0
        if producer is not None and fused.KNN_PREP:
            lk = (groups, L, plan.k, plan.d, key.fused_mr)                 # the Grapher in front prepares this graph's keys
            if producer.__dict__.get("_gkg_label_knn") != lk:              # from its next call on (fused.grapher_label_forward)
                producer._gkg_label_knn = lk
        ctx.save_for_backward(buf, e2, ft, w1, wc, w2, w4, w5)
        ctx.desc = d
        ctx.plan = plan
        ctx.mark_non_differentiable(edge)
        ctx.set_materialize_grads(False)
        return out, edge

    @staticmethod
    def backward(ctx, dout, _gedge=None):
        lib = _lib.load()
        if dout is None:
            return (None,) * 20
        plan = ctx.plan
        B, Cc, L, M, Cf = plan.dims
        T = B * L
        buf, e2, ft, w1, wc, w2, w4, w5 = ctx.saved_tensors
        dev = buf.device
        d = ctx.desc
        tbuf = torch.empty(plan.bwd_total, dtype=_F32, device=dev)
        base = tbuf.data_ptr()
        d.dY5, d.dh2, d.dY3, d.gx1, d.dY1, d.df1, d.dY4, d.dA2, d.dY2, d.dXM = [base + 4 * o for o in plan.bwd_offs]
        de = torch.empty((T, Cc), dtype=_F32, device=dev)
        dft = torch.empty((B, M, Cc), dtype=_F32, device=dev)
        dout_c = dout.contiguous()
        d.dout, d.de, d.dft = dout_c.data_ptr(), de.data_ptr(), dft.data_ptr()
        scratch = fused._BnBwdScratch.of(dev)
        wq = (_lib.WgradProblem * 5)()
        p1, pc, p3, p4, p5 = plan.projs
        scratch.hold = 1                 # every layer's buffers are handed out before the first launch (bn_scratch.one_call)
        try:
            o5 = _proj_bwd(lib, d.ffn2, p5, scratch, dev)
            o4 = _proj_bwd(lib, d.ffn1, p4, scratch, dev)
            o3 = _proj_bwd(lib, d.fc2, p3, scratch, dev)
            oc = _proj_bwd(lib, d.conv, pc, scratch, dev)
            o1 = _proj_bwd(lib, d.fc1, p1, scratch, dev)
            _lib.check(lib.gkg_grapher_label_bwd(C.byref(d), wq, _stream()), "gkg_grapher_label_bwd")
        except Exception:
            scratch.poison()
            raise
        finally:
            scratch.hold = 0
        _issue_wgrads(lib, wq, 5, (o5[0], o4[0], o3[0], oc[0], o1[0]), (buf, tbuf, e2, dout_c), dev)
        return (de, dft, o1[0].view_as(w1), o1[1], o1[2], oc[0].view_as(wc), oc[1], oc[2], o3[0].view_as(w2), o3[1], o3[2],
                o4[0].view_as(w4), o4[1], o4[2], o5[0].view_as(w5), o5[1], o5[2], None, None, None)


def label_forward(mod, e2, ft, groups, keys_key, producer=None):
    """After label_ok(): the block through the driver (fused.grapher_label_forward; builds the plan the next calls go through)."""
    plan = _plan_label(mod, e2, ft, groups)
    return _LabelBlockFn.apply(e2, ft, *plan.params, plan, keys_key, producer)


def try_label(mod, e, features):
    """GrapherLabel.forward's first question (see try_grapher): -> (E', edge_index) or None."""
    plans = _PLANS.get(mod)
    if plans is None or not (ENABLED and e.is_cuda and e.dtype == _F32 and e.dim() == 3 and e.is_contiguous() and features.is_cuda
                             and features.dtype == _F32 and features.dim() == 4 and torch.is_grad_enabled()
                             and not torch.is_autocast_enabled()):
        return None
    B, L, Cc = e.shape
    if features.shape[0] != B or features.shape[1] != Cc or fused.is_channels_last(features):
        return None
    key = ("l", B * L, (B, features.shape[2] * features.shape[3], Cc), mod.graph_conv.num_head, _switches())
    plan = plans.get(key)
    if plan is None or not plan.valid():
        return None
    ft, keys_key, producer = fused._label_features(features, B, Cc)
    out, edge = _LabelBlockFn.apply(e.view(B * L, Cc), ft, *plan.params, plan, keys_key, producer)
    return out.view(B, L, Cc), edge
