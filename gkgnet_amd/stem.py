"""Autograd Functions of the backbone's stem / downsample path (reference gkgnet.py:74-118): the first stem convolution on the
direct kernel (csrc/gkg_stem.hip), library convolutions in front of a train-mode BN with a lean backward, and the pos_embed
add in the feature map's own layout.  The policy (when each is used) lives in fused.py."""
from __future__ import annotations

import torch

from . import _lib
from .ops import _ptr, _stream

_F32 = torch.float32


class _StemConv(torch.autograd.Function):
    """y = conv(x) + bias as a channels-last (B, cout, Ho, Wo) tensor; the weight / bias gradients come from the library's
    convolution backward (the image itself needs no gradient in the backbone; it is computed when asked for)."""

    @staticmethod
    def forward(ctx, x, weight, bias, bn_behind=False):
        lib = _lib.load()
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        ctx.bn_behind = bool(bn_behind)
        x = x.contiguous()
        out = torch.empty((B, (H + 1) // 2, (W + 1) // 2, cout), dtype=_F32, device=x.device)
        _lib.check(lib.gkg_stem_conv3x3s2_fwd(_ptr(x), _ptr(weight.contiguous()), _ptr(bias), None, None, _ptr(out), B, cin, H, W, cout,
                                              0, _lib.F32, _stream()), "gkg_stem_conv3x3s2_fwd")
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return out.permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        # the gradient arrives channels-last (the BN kernels behind the convolution are token-major): the image is laid out the
        # same way for the library's backward (127 MB at GKGNet-576 / B = 32) instead of the library copying the gradient
        # (425 MB, 1.0 ms) to the image's layout
        if g.is_contiguous(memory_format=torch.channels_last) and not g.is_contiguous():
            x = x.contiguous(memory_format=torch.channels_last)
        bias_grad = ctx.has_bias and ctx.needs_input_grad[2] and not ctx.bn_behind
        mask = [ctx.needs_input_grad[0], ctx.needs_input_grad[1], bias_grad]
        gx, gw, gb = torch.ops.aten.convolution_backward(g, x, weight, [weight.shape[0]] if ctx.has_bias else None, [2, 2], [1, 1],
                                                         [1, 1], False, [0, 0], 1, mask)
        if ctx.has_bias and ctx.needs_input_grad[2] and ctx.bn_behind:
            gb = torch.zeros(weight.shape[0], dtype=g.dtype, device=g.device)     # exactly zero behind a train-mode BN (_ConvBeforeBN)
        return gx, gw, gb, None


class _ConvBeforeBN(torch.autograd.Function):
    """A library convolution whose output goes straight into a TRAIN-mode BatchNorm: channels-last operands on both sides of
    the backward (no layout copy of the incoming gradient), and no bias-gradient reduction — the BN removes the per-channel
    mean, so the bias gradient is exactly zero in exact arithmetic (the library spends a 0.6 TB/s reduction over the whole
    output on it: 340 us per stem convolution at GKGNet-576, B = 32); a zero tensor is returned for it."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding):
        xc = x.contiguous(memory_format=torch.channels_last)
        y = torch.ops.aten.convolution(xc, weight, bias, stride, padding, [1, 1], False, [0, 0], 1)
        ctx.save_for_backward(xc, weight)
        ctx.conf = (stride, padding, None if bias is None else bias.shape[0])
        return y

    @staticmethod
    def backward(ctx, g):
        xc, weight = ctx.saved_tensors
        stride, padding, nbias = ctx.conf
        g = g.contiguous(memory_format=torch.channels_last)
        gx, gw, _ = torch.ops.aten.convolution_backward(g, xc, weight, None, stride, padding, [1, 1], False, [0, 0], 1,
                                                        [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        gb = None if nbias is None or not ctx.needs_input_grad[2] else torch.zeros(nbias, dtype=g.dtype, device=g.device)
        return gx, gw, gb, None, None


class _AddPosEmbed(torch.autograd.Function):
    """x (B, C, H, W) channels-last + pos_embed (1, C, H, W): the broadcast add against an NCHW parameter runs at 0.76 TB/s
    through the generic strided kernel and its backward (sum over the batch of a channels-last gradient) at 0.63 TB/s; with
    the parameter laid out like x both are plain streaming passes."""

    @staticmethod
    def forward(ctx, x, pos):
        ctx.pos_shape = pos.shape
        return x + pos.contiguous(memory_format=torch.channels_last)

    @staticmethod
    def backward(ctx, g):
        gp = None
        if ctx.needs_input_grad[1]:
            B, C, H, W = g.shape
            gl = g.permute(0, 2, 3, 1)                                   # channels-last gradient: a contiguous (B, H*W*C) matrix
            if not gl.is_contiguous():
                gl = gl.contiguous()
            gp = gl.reshape(B, -1).sum(0).view(1, H, W, C).permute(0, 3, 1, 2)
        return g, gp
