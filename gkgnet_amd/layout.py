"""Layout changes at the block boundary as autograd Functions: (B, C, *spatial) <-> token-major (B*N, C) through the layout
kernels (gkg_nchw_to_tm / gkg_tm_affine_to_nchw), and the zero-cost channels-last view of a token-major matrix."""
from __future__ import annotations

import torch

from . import _lib
from .ops import _ptr, _stream

_F32 = torch.float32


def _tm_dtype(lowp: bool):
    return (torch.bfloat16, _lib.BF16) if lowp else (_F32, _lib.F32)


class _TokenMajorToCL(torch.autograd.Function):
    """(B*H*W, C) token-major -> logical (B, C, H, W) in channels-last memory (a view).  The backward accepts either
    memory format: a channels-last gradient is a view again, an NCHW one goes through the layout kernel."""

    @staticmethod
    def forward(ctx, t, B, H, W):
        ctx.dims = (B, H, W)
        return t.view(B, H, W, t.shape[1]).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        B, H, W = ctx.dims
        C = g.shape[1]
        if g.dtype == _F32 and g.permute(0, 2, 3, 1).is_contiguous():
            return g.permute(0, 2, 3, 1).reshape(B * H * W, C), None, None, None
        g = g.float().contiguous()
        out = torch.empty((B * H * W, C), dtype=_F32, device=g.device)
        _lib.check(_lib.load().gkg_nchw_to_tm(_ptr(g), _ptr(out), B, C, H * W, _lib.F32, None, _stream()), "gkg_nchw_to_tm")
        return out, None, None, None


class _ToTokenMajor(torch.autograd.Function):
    """(B, C, *spatial) -> (B*N, C)."""

    @staticmethod
    def forward(ctx, x):
        B, C = x.shape[:2]
        N = x[0, 0].numel()
        x = x.contiguous()
        out = torch.empty((B * N, C), dtype=_F32, device=x.device)
        _lib.check(_lib.load().gkg_nchw_to_tm(_ptr(x), _ptr(out), B, C, N, _lib.F32, None, _stream()), "gkg_nchw_to_tm")
        ctx.shape = tuple(x.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C = ctx.shape[:2]
        N = g.shape[0] // B
        g = g.contiguous()
        out = torch.empty(ctx.shape, dtype=_F32, device=g.device)
        _lib.check(_lib.load().gkg_tm_affine_to_nchw(_ptr(g), None, None, None, _ptr(out), B, C, N, None, _stream()),
                   "gkg_tm_affine_to_nchw")
        return out


class _BlockEntry(torch.autograd.Function):
    """x (B, C, *spatial) -> (x_tm (B*N, C), x): a block's token-major input and its residual branch leave ONE
    autograd node, so the backward receives both incoming gradients together and adds them inside the layout
    kernel (instead of a separate layout pass followed by autograd's elementwise accumulation)."""

    @staticmethod
    def forward(ctx, x, lowp=False):
        B, C = x.shape[:2]
        N = x[0, 0].numel()
        dt, code = _tm_dtype(lowp)
        out = torch.empty((B * N, C), dtype=dt, device=x.device)
        _lib.check(_lib.load().gkg_nchw_to_tm(_ptr(x), _ptr(out), B, C, N, code, None, _stream()), "gkg_nchw_to_tm")
        ctx.shape = tuple(x.shape)
        ctx.set_materialize_grads(False)
        return out, x.view_as(x)

    @staticmethod
    def backward(ctx, g_tm, g_res):
        if g_tm is None:
            return g_res, None
        B, C = ctx.shape[:2]
        N = g_tm.shape[0] // B
        res = None if g_res is None else g_res.contiguous()
        out = torch.empty(ctx.shape, dtype=_F32, device=g_tm.device)
        g_tm = g_tm.contiguous()                 # named: the copy must outlive the launch that reads it
        _lib.check(_lib.load().gkg_tm_affine_to_nchw(_ptr(g_tm), None, None, _ptr(res), _ptr(out), B, C, N, None,
                                                     _stream()), "gkg_tm_affine_to_nchw")
        return out, None


class _AvgPoolTM(torch.autograd.Function):
    """avg_pool2d(r, r) of a token-major feature map (the pooled key set of the reference, torch_vertex.py:194-196):
    x (B, H W, C) — or the x half of an XM operand buffer, (B, H W, 4, C/4) strided (fused._xm_xview) — -> (B, (H/r)(W/r), C).
    Forward: gkg_avgpool_tm (reads the view in place).  Backward: every pooled gradient goes to its r x r window divided by
    r^2 — one broadcast copy; the library's NHWC backward ran at 0.87 TB/s (243 us per GKGNet-576 stage-1 block, B = 32)."""

    @staticmethod
    def forward(ctx, x, H, W, r):
        from .fused import _tm_view, _as_tokens
        B, N, C, ldx, xchunk = _tm_view(x)
        x = _as_tokens(x)
        ctx.r, ctx.hw, ctx.xshape = r, (H, W), tuple(x.shape)
        y = torch.empty((B, (H // r) * (W // r), C), dtype=_F32, device=x.device)
        _lib.check(_lib.load().gkg_avgpool_tm(_ptr(x), ldx, xchunk, _ptr(y), B, H, W, C, r, _stream()), "gkg_avgpool_tm")
        return y

    @staticmethod
    def backward(ctx, g):
        r = ctx.r
        H, W = ctx.hw
        Hr, Wr = H // r, W // r
        B, C = g.shape[0], g.shape[2]
        g = g.reshape(B, Hr, Wr, C)
        gs = (g * (1.0 / (r * r))).view(B, Hr, 1, Wr, 1, C).expand(B, Hr, r, Wr, r, C)
        if Hr * r == H and Wr * r == W:
            return gs.reshape(ctx.xshape), None, None, None
        out = g.new_zeros((B, H, W, C))                 # floor mode: rows / columns past the last full window get no gradient
        out[:, :Hr * r, :Wr * r] = gs.reshape(B, Hr * r, Wr * r, C)
        return out.view(ctx.xshape), None, None, None
