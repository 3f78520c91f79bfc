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

import torch

from . import _lib, fused
from .ops import _ptr, _stream
from .parallel import grad_view

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


# ----------------------------------------------------------------------------------------------- descriptor pieces
def _fill_proj(lib, p: ProjBN, seq, nb, cin, cout, kperm, scratch, keep):
    """Forward half of a projection's descriptor: planes, BN parameters, the forward BN pass's scratch buffers."""
    conv, bn = seq[0], seq[1]
    pf, pd = fused._planes(lib, conv.weight, nb, cout, cin, True, True, kperm=kperm)
    track = bn.training and bn.track_running_stats
    fused._touch_stats(bn, track)
    cur, other, zero = scratch.acquire(lib, 2 * nb * cout)
    p.planes_fwd, p.planes_dgrad = _ptr(pf), _ptr(pd)
    p.gamma, p.beta, p.bias = _ptr(bn.weight), _ptr(bn.bias), _ptr(conv.bias)
    p.running_mean = _ptr(bn.running_mean) if track else None
    p.running_var = _ptr(bn.running_var) if track else None
    p.nbt = _ptr(bn.num_batches_tracked) if track else None
    p.momentum, p.eps = float(bn.momentum), float(bn.eps)
    p.cin, p.cout, p.nb = cin, cout, nb
    p.fsum, p.fzero, p.fzero_n = _ptr(cur), _ptr(other), zero
    keep.append((pf, pd))


def _fill_proj_bwd(lib, p: ProjBN, params, wshape, nch, scratch, dev):
    """Backward half: the BN pass's scratch buffers and the gradient outputs (bucket slots when the parameters have them)."""
    dWv, dgamma, dbeta = fused._grad_outs(params, wshape, nch, dev)
    if not getattr(dWv, "_gkg_zero", False):
        dWv.zero_()                                      # the weight-gradient kernels ADD into dw
    cur, other, zero = scratch.acquire(lib, 2 * nch)
    p.bsum, p.bzero, p.bzero_n = _ptr(cur), _ptr(other), zero
    p.dw, p.dgamma, p.dbeta = _ptr(dWv), _ptr(dgamma), _ptr(dbeta)
    return dWv, dgamma, dbeta


class _Arena:
    """One allocation, carved into fp32 tensors (16-byte aligned pieces)."""

    def __init__(self, device):
        self.device, self.sizes = device, []

    def add(self, *shape):
        n = 1
        for s in shape:
            n *= s
        self.sizes.append((shape, (n + 3) & ~3))
        return len(self.sizes) - 1

    def build(self):
        buf = torch.empty(sum(n for _, n in self.sizes), dtype=_F32, device=self.device)
        out, o = [], 0
        for shape, n in self.sizes:
            m = 1
            for s in shape:
                m *= s
            out.append(buf[o:o + m].view(shape))
            o += n
        return buf, out


def _graph_op(lib, g: GraphOp, x_like, B, G, c, N, M, k, d, relative_pos, has_y, want_edge, nn_, dev, keys_key, keep):
    """The block's k-NN + aggregation: kernel form, flags, workspace (shared with a keys producer when the Grapher in front prepared
    this graph's keys).  -> (key object for the prepared queries, edge tensor | None)."""
    C_ = G * c
    fm = fused._knn_mr_shapes_ok(B, N, C_, M, has_y, relative_pos, k, d, G, nn_, False)
    key = fused._KnnKey(B, G, c, N, M, k, d, has_y, relative_pos, fm)
    flags = key.flags
    if fused.KNN_PREP:
        flags |= _lib.KNN_X_PREPARED
        if keys_key is not None and keys_key.ws is not None and keys_key.tuple() == key.tuple():
            key.ws, key.y_ready = keys_key.ws, True
            flags |= _lib.KNN_Y_PREPARED
    if key.ws is None:
        key.ws = fused._ws(lib.gkg_knn_workspace_bytes(B * G, c, N, M, k, d, _lib.F32, _lib.KNN_NORMALIZE), dev)
    rp = None
    if relative_pos is not None:
        rp = fused._rp_arg(relative_pos, N, M)
    g.G, g.k, g.d, g.fused_mr = G, k, d, int(fm)
    g.relpos, g.knn_flags, g.mr_flags = _ptr(rp), flags, fused._mr_bwd_flags()
    g.knn_ws, g.knn_ws_bytes = _ptr(key.ws), key.ws.numel()
    edge = None
    g.nn16 = g.nn_idx = g.center = None
    if want_edge:
        edge = torch.empty((2, B * G, N, k), dtype=torch.int64, device=dev)
        g.nn_idx, g.center = edge[0].data_ptr(), edge[1].data_ptr()
    elif not fm:
        nn16 = torch.empty((B * G, N, k), dtype=torch.int16, device=dev)
        g.nn16 = _ptr(nn16)
        keep.append(nn16)
    keep.append((rp, key.ws))
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
    def forward(ctx, x, w1, g1, b1, wc, gc_, bc, w2, g2, b2, mod, relative_pos, groups, dual):
        lib = _lib.load()
        _bind(lib)
        gcv = mod.graph_conv
        nn_ = gcv.gconv.nn
        B, Cc, H, W = x.shape
        N, T, dev = H * W, B * H * W, x.device
        x = x.contiguous()
        d = GrapherBlock()
        keep = []
        scratch = fused._BnFwdScratch.of(dev)
        ar = _Arena(dev)
        ixt, iXM, iA2 = ar.add(T, Cc), ar.add(T, 2 * Cc), ar.add(T, 2 * Cc)
        iY1, iY2, iY3 = ar.add(T, Cc), ar.add(4, T, Cc // 2), ar.add(T, Cc)
        ib1, ib2, ib3 = ar.add(4, Cc), ar.add(4, 2 * Cc), ar.add(4, Cc)
        iarg = ar.add(T, Cc // 2)                               # (T, C) u16
        buf, t = ar.build()
        out = torch.empty((B, Cc, H, W), dtype=_F32, device=dev)
        out_tm = torch.empty((T, Cc), dtype=_F32, device=dev) if dual else None
        d.B, d.C, d.H, d.W = B, Cc, H, W
        d.x, d.out, d.out_tm = _ptr(x), _ptr(out), _ptr(out_tm)
        d.xt, d.XM, d.A2 = _ptr(t[ixt]), _ptr(t[iXM]), _ptr(t[iA2])
        scratch.hold = 1                 # every layer's buffers are handed out before the first launch (bn_scratch.one_call)
        try:
            _fill_proj(lib, d.fc1, mod.fc1, 1, Cc, Cc, 0, scratch, keep)
            _fill_proj(lib, d.conv, nn_, 4, Cc // 2, Cc // 2, 1, scratch, keep)
            _fill_proj(lib, d.fc2, mod.fc2, 1, 2 * Cc, Cc, 0, scratch, keep)
            d.fc1.Y, d.fc1.bn = _ptr(t[iY1]), _ptr(t[ib1])
            d.conv.Y, d.conv.bn = _ptr(t[iY2]), _ptr(t[ib2])
            d.fc2.Y, d.fc2.bn = _ptr(t[iY3]), _ptr(t[ib3])
            _graph_op(lib, d.graph, x, B, groups, Cc // groups, N, N, gcv.k, gcv.d, relative_pos, False, False, nn_, dev, None, keep)
            d.graph.arg = _ptr(t[iarg])
            sk = fused._sk_ws(dev)
            d.sk_ws, d.sk_bytes = _ptr(sk), sk.numel()
            kk = None
            lk = getattr(mod, "_gkg_label_knn", None) if (dual and fused.KNN_PREP) else None
            if lk is not None:
                G2, L2, k2, d2, fm2 = lk
                if Cc % G2 == 0 and (Cc // G2) % 4 == 0:
                    kk = fused._KnnKey(B, G2, Cc // G2, L2, N, k2, d2, True, None, fm2)
                    kk.as_keys = 1
                    kk.ws = fused._ws(lib.gkg_knn_workspace_bytes(B * G2, Cc // G2, L2, N, k2, d2, _lib.F32, _lib.KNN_NORMALIZE), dev)
                    d.keys_G, d.keys_L, d.keys_k, d.keys_d, d.keys_fused_mr, d.keys_flags = G2, L2, k2, d2, int(fm2), kk.flags
                    d.keys_ws, d.keys_ws_bytes = _ptr(kk.ws), kk.ws.numel()
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
        ctx.params = ((w1, g1, b1), (wc, gc_, bc), (w2, g2, b2))
        ctx.dims = (B, Cc, H, W, dual)
        if dual:
            ctx.set_materialize_grads(False)
            return out, out_tm
        return out

    @staticmethod
    def backward(ctx, dout, dtm=None):
        lib = _lib.load()
        B, Cc, H, W, dual = ctx.dims
        if dout is None and dtm is None:
            return (None,) * 14
        T, dev = B * H * W, ctx.saved_tensors[0].device
        d = ctx.desc
        ar = _Arena(dev)
        names = [ar.add(T, Cc) for _ in range(5)] + [ar.add(T, 2 * Cc) for _ in range(3)]
        buf, t = ar.build()
        g3, dY3, gx1, dY1, dxt, dA2, dY2, dXM = t
        dx = torch.empty((B, Cc, H, W), dtype=_F32, device=dev)
        if dout is None:                                          # only the token-major companion was used downstream
            dout = torch.zeros((B, Cc, H, W), dtype=_F32, device=dev)
        dout_c = dout.contiguous()
        dtm_c = None if dtm is None else dtm.contiguous()
        d.dout, d.dout_tm, d.dx = _ptr(dout_c), _ptr(dtm_c), _ptr(dx)
        d.g3, d.dY3, d.dA2, d.dY2, d.dXM, d.gx1, d.dY1, d.dxt = (_ptr(g3), _ptr(dY3), _ptr(dA2), _ptr(dY2), _ptr(dXM), _ptr(gx1), _ptr(dY1),
                                                                 _ptr(dxt))
        scratch = fused._BnBwdScratch.of(dev)
        wq = (_lib.WgradProblem * 3)()
        scratch.hold = 1                 # every layer's buffers are handed out before the first launch (bn_scratch.one_call)
        try:
            o2 = _fill_proj_bwd(lib, d.fc2, ctx.params[2], (Cc, 2 * Cc), Cc, scratch, dev)
            oc = _fill_proj_bwd(lib, d.conv, ctx.params[1], (4, Cc // 2, Cc // 2), 2 * Cc, scratch, dev)
            o1 = _fill_proj_bwd(lib, d.fc1, ctx.params[0], (Cc, Cc), Cc, scratch, dev)
            _lib.check(lib.gkg_grapher_bwd(C.byref(d), wq, _stream()), "gkg_grapher_bwd")
        except Exception:
            scratch.poison()
            raise
        finally:
            scratch.hold = 0
        _issue_wgrads(lib, wq, 3, (o2[0], oc[0], o1[0]), (buf, ctx.saved_tensors[0], dout_c, dtm_c), dev)
        w1, wc, w2 = ctx.saved_tensors[1:]
        return (dx, o1[0].view_as(w1), o1[1], o1[2], oc[0].view_as(wc), oc[1], oc[2], o2[0].view_as(w2), o2[1], o2[2], None, None, None, None)


def grapher_forward(mod, x, relative_pos, groups, dual):
    nn_ = mod.graph_conv.gconv.nn
    res = _GrapherBlockFn.apply(x, mod.fc1[0].weight, mod.fc1[1].weight, mod.fc1[1].bias, nn_[0].weight, nn_[1].weight, nn_[1].bias,
                                mod.fc2[0].weight, mod.fc2[1].weight, mod.fc2[1].bias, mod, relative_pos, groups, dual)
    return res


# ----------------------------------------------------------------------------------------------- GrapherLabel
class _LabelBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e2, ft, w1, g1, b1, wc, gc_, bc, w2, g2, b2, w4, g4, b4, w5, g5, b5, mod, groups, keys_key):
        lib = _lib.load()
        _bind(lib)
        gcv = mod.graph_conv
        nn_ = gcv.gconv.nn
        B, M, Cc = ft.shape
        T = e2.shape[0]
        L, dev = T // B, e2.device
        Cf = w4.shape[0]
        d = LabelBlock()
        keep = []
        scratch = fused._BnFwdScratch.of(dev)
        ar = _Arena(dev)
        iXM, iA2, ih2, if1 = ar.add(T, 2 * Cc), ar.add(T, 2 * Cc), ar.add(T, Cc), ar.add(T, Cf)
        iY1, iY2, iY3, iY4, iY5 = ar.add(T, Cc), ar.add(4, T, Cc // 2), ar.add(T, Cc), ar.add(T, Cf), ar.add(T, Cc)
        ib1, ib2, ib3, ib4, ib5 = ar.add(4, Cc), ar.add(4, 2 * Cc), ar.add(4, Cc), ar.add(4, Cf), ar.add(4, Cc)
        iarg = ar.add(T, Cc // 2)
        buf, t = ar.build()
        out = torch.empty((T, Cc), dtype=_F32, device=dev)
        d.B, d.C, d.L, d.M = B, Cc, L, M
        d.e, d.ft, d.out = _ptr(e2), _ptr(ft), _ptr(out)
        d.XM, d.A2, d.h2, d.f1 = _ptr(t[iXM]), _ptr(t[iA2]), _ptr(t[ih2]), _ptr(t[if1])
        scratch.hold = 1                 # every layer's buffers are handed out before the first launch (bn_scratch.one_call)
        try:
            _fill_proj(lib, d.fc1, mod.fc1, 1, Cc, Cc, 0, scratch, keep)
            _fill_proj(lib, d.conv, nn_, 4, Cc // 2, Cc // 2, 1, scratch, keep)
            _fill_proj(lib, d.fc2, mod.fc2, 1, 2 * Cc, Cc, 0, scratch, keep)
            _fill_proj(lib, d.ffn1, mod.ffn.fc1, 1, Cc, Cf, 0, scratch, keep)
            _fill_proj(lib, d.ffn2, mod.ffn.fc2, 1, Cf, Cc, 0, scratch, keep)
            for p, iy, ib in ((d.fc1, iY1, ib1), (d.conv, iY2, ib2), (d.fc2, iY3, ib3), (d.ffn1, iY4, ib4), (d.ffn2, iY5, ib5)):
                p.Y, p.bn = _ptr(t[iy]), _ptr(t[ib])
            _, edge = _graph_op(lib, d.graph, e2, B, groups, Cc // groups, L, M, gcv.k, gcv.d, None, True, True, nn_, dev, keys_key, keep)
            d.graph.arg = _ptr(t[iarg])
            sk = fused._sk_ws(dev)
            d.sk_ws, d.sk_bytes = _ptr(sk), sk.numel()
            _lib.check(lib.gkg_grapher_label_fwd(C.byref(d), _stream()), "gkg_grapher_label_fwd")
        except Exception:
            scratch.poison()
            raise
        finally:
            scratch.hold = 0
        ctx.save_for_backward(buf, e2, ft, w1, wc, w2, w4, w5)
        ctx.desc = d
        ctx.params = ((w1, g1, b1), (wc, gc_, bc), (w2, g2, b2), (w4, g4, b4), (w5, g5, b5))
        ctx.dims = (B, Cc, L, M, Cf)
        ctx.mark_non_differentiable(edge)
        ctx.set_materialize_grads(False)
        return out, edge

    @staticmethod
    def backward(ctx, dout, _gedge=None):
        lib = _lib.load()
        if dout is None:
            return (None,) * 20
        B, Cc, L, M, Cf = ctx.dims
        T = B * L
        buf, e2, ft, w1, wc, w2, w4, w5 = ctx.saved_tensors
        dev = buf.device
        d = ctx.desc
        ar = _Arena(dev)
        ids = [ar.add(T, Cc) for _ in range(5)] + [ar.add(T, Cf) for _ in range(2)] + [ar.add(T, 2 * Cc) for _ in range(3)]
        tbuf, t = ar.build()
        dY5, dh2, dY3, gx1, dY1, df1, dY4, dA2, dY2, dXM = t
        de = torch.empty((T, Cc), dtype=_F32, device=dev)
        dft = torch.empty((B, M, Cc), dtype=_F32, device=dev)
        dout_c = dout.contiguous()
        d.dout, d.de, d.dft = _ptr(dout_c), _ptr(de), _ptr(dft)
        d.dY5, d.df1, d.dY4, d.dh2, d.dY3, d.dA2, d.dY2, d.dXM, d.gx1, d.dY1 = (_ptr(dY5), _ptr(df1), _ptr(dY4), _ptr(dh2), _ptr(dY3), _ptr(dA2),
                                                                                   _ptr(dY2), _ptr(dXM), _ptr(gx1), _ptr(dY1))
        scratch = fused._BnBwdScratch.of(dev)
        wq = (_lib.WgradProblem * 5)()
        scratch.hold = 1                 # every layer's buffers are handed out before the first launch (bn_scratch.one_call)
        try:
            o5 = _fill_proj_bwd(lib, d.ffn2, ctx.params[4], (Cc, Cf), Cc, scratch, dev)
            o4 = _fill_proj_bwd(lib, d.ffn1, ctx.params[3], (Cf, Cc), Cf, scratch, dev)
            o3 = _fill_proj_bwd(lib, d.fc2, ctx.params[2], (Cc, 2 * Cc), Cc, scratch, dev)
            oc = _fill_proj_bwd(lib, d.conv, ctx.params[1], (4, Cc // 2, Cc // 2), 2 * Cc, scratch, dev)
            o1 = _fill_proj_bwd(lib, d.fc1, ctx.params[0], (Cc, Cc), Cc, scratch, dev)
            _lib.check(lib.gkg_grapher_label_bwd(C.byref(d), wq, _stream()), "gkg_grapher_label_bwd")
        except Exception:
            scratch.poison()
            raise
        finally:
            scratch.hold = 0
        _issue_wgrads(lib, wq, 5, (o5[0], o4[0], o3[0], oc[0], o1[0]), (buf, tbuf, e2, dout_c), dev)
        return (de, dft, o1[0].view_as(w1), o1[1], o1[2], oc[0].view_as(wc), oc[1], oc[2], o3[0].view_as(w2), o3[1], o3[2],
                o4[0].view_as(w4), o4[1], o4[2], o5[0].view_as(w5), o5[1], o5[2], None, None, None)


def label_forward(mod, e2, ft, groups, keys_key):
    nn_ = mod.graph_conv.gconv.nn
    return _LabelBlockFn.apply(e2, ft, mod.fc1[0].weight, mod.fc1[1].weight, mod.fc1[1].bias, nn_[0].weight, nn_[1].weight, nn_[1].bias,
                               mod.fc2[0].weight, mod.fc2[1].weight, mod.fc2[1].bias, mod.ffn.fc1[0].weight, mod.ffn.fc1[1].weight,
                               mod.ffn.fc1[1].bias, mod.ffn.fc2[0].weight, mod.ffn.fc2[1].weight, mod.ffn.fc2[1].bias, mod, groups,
                               keys_key)
