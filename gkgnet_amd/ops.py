"""PyTorch-facing operators of the Group-KNN hot path, backed by libgkg_hip.so (HIP, gfx950).

    knn_graph(x, y, relative_pos, k, dilation)  ==  DenseDilatedKnnGraph.forward   (reference torch_edge.py:164-176)
    max_relative(x, nn_idx, y)                  ==  max_k(gather(y|x, idx) - x)    (reference torch_vertex.py:49-54)

Tensors keep the reference layout (B*G, c, N, 1) / (B*G, c, N).  torch is used only for device
memory and the current HIP stream; all arithmetic runs in the library.  There is no CPU path:
CPU tensors raise.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}

def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.GkgError("gkgnet_amd ops run on the GPU only (HIP kernels); got a CPU tensor — "
                                "there is no CPU fallback")


def _ptr(t):
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current device's current stream as the integer the C-ABI takes.  (torch.cuda.current_stream() builds a Stream object
    through three Python layers — ~8 us a call, 27 calls in an eager cfg2 forward; the raw query is one C call.)"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _tokens(t: torch.Tensor) -> torch.Tensor:
    """(BG,c,N,1) or (BG,c,N) -> contiguous (BG,c,N)."""
    if t.dim() == 4:
        t = t.reshape(t.shape[0], t.shape[1], -1)
    return t.contiguous()


@torch.no_grad()
def knn_graph(x: torch.Tensor, y: Optional[torch.Tensor] = None, relative_pos: Optional[torch.Tensor] = None,
              k: int = 9, dilation: int = 1, normalize: bool = True, want_center: bool = True) -> torch.Tensor:
    """Dilated k-NN graph.  Returns edge_index (2, BG, N, k) int64 ([0] neighbours sorted by ascending
    distance, every ``dilation``-th of the top k*dilation kept; [1] centre index), like the reference."""
    _need_cuda(x, y, relative_pos)
    lib = _lib.load()
    xq = _tokens(x.detach())
    yk = None if y is None else _tokens(y.detach())
    if xq.dtype not in _DT:
        raise _lib.GkgError(f"unsupported dtype {xq.dtype} (fp32 / bf16 / fp16)")
    if yk is not None and (yk.dtype != xq.dtype or yk.shape[:2] != xq.shape[:2]):
        raise _lib.GkgError("y must match x in dtype, batch*groups and channels")
    BG, c, N = xq.shape
    M = N if yk is None else yk.shape[2]
    rp = None
    if relative_pos is not None:
        rp = relative_pos.detach().to(torch.float32).reshape(-1, relative_pos.shape[-1]).contiguous()
        if tuple(rp.shape) != (N, M):
            raise _lib.GkgError(f"relative_pos must be (1,{N},{M}), got {tuple(relative_pos.shape)}")
    flags = (_lib.KNN_NORMALIZE if normalize else 0) | _lib.knn_select_flags() | _lib.relpos_flags(relative_pos)
    dt = _DT[xq.dtype]
    edge = torch.empty((2 if want_center else 1, BG, N, k), dtype=torch.int64, device=xq.device)
    nbytes = lib.gkg_knn_workspace_bytes(BG, c, N, M, k, dilation, dt, flags)
    # nbytes == 0 means the sizes are unsupported: pass a token buffer and let the call produce the diagnostic
    ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=xq.device)
    rc = lib.gkg_knn_fwd(_ptr(xq), _ptr(yk), _ptr(rp), edge[0].data_ptr(),
                         edge[1].data_ptr() if want_center else None, BG, c, N, M, k, dilation, dt, flags,
                         ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "gkg_knn_fwd")
    return edge


class _MaxRelative(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, src, nn_idx):
        lib = _lib.load()
        BG, c, N = x.shape
        M = N if src is None else src.shape[2]
        k = nn_idx.shape[2]
        m = torch.empty_like(x)
        need_grad = any(ctx.needs_input_grad[:2])
        arg = torch.empty((BG, c, N), dtype=torch.uint8, device=x.device) if need_grad else None
        rc = lib.gkg_mr_fwd(_ptr(x), _ptr(src), _ptr(nn_idx), _ptr(m), _ptr(arg), BG, c, N, M, k, _DT[x.dtype],
                            _stream())
        _lib.check(rc, "gkg_mr_fwd")
        ctx.save_for_backward(nn_idx, arg)
        ctx.dims = (BG, c, N, M, k, src is not None, x.dtype)
        return m

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        nn_idx, arg = ctx.saved_tensors
        BG, c, N, M, k, has_src, dtype = ctx.dims
        g = g.contiguous()
        gx = torch.empty((BG, c, N), dtype=dtype, device=g.device)
        gsrc = torch.empty((BG, c, M), dtype=dtype, device=g.device) if has_src else None
        rc = lib.gkg_mr_bwd(_ptr(g), _ptr(nn_idx), _ptr(arg), _ptr(gx), _ptr(gsrc), BG, c, N, M, k, _DT[dtype],
                            _stream())
        _lib.check(rc, "gkg_mr_bwd")
        return gx, gsrc, None


def max_relative(x: torch.Tensor, nn_idx: torch.Tensor, y: Optional[torch.Tensor] = None) -> torch.Tensor:
    """m[b,ch,n] = max_j (src[b,ch,nn_idx[b,n,j]] - x[b,ch,n]), src = y if given else x.  Differentiable
    w.r.t. x and y (the indices are constants, as in the reference).  x (BG,c,N[,1]) -> (BG,c,N)."""
    _need_cuda(x, y, nn_idx)
    xs = _tokens(x)
    ys = None if y is None else _tokens(y)
    if xs.dtype not in _DT:
        raise _lib.GkgError(f"unsupported dtype {xs.dtype} (fp32 / bf16 / fp16)")
    if ys is not None and ys.dtype != xs.dtype:
        raise _lib.GkgError("y must match x in dtype")
    idx = nn_idx.contiguous()
    if idx.dtype != torch.int64 or idx.dim() != 3 or tuple(idx.shape[:2]) != (xs.shape[0], xs.shape[2]):
        raise _lib.GkgError(f"nn_idx must be int64 (BG,N,k); got {idx.dtype} {tuple(idx.shape)}")
    return _MaxRelative.apply(xs, ys, idx)


# ----------------------------------------------------------------------------------------------- EdgeConv aggregation
ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2


class _EdgeAggregate(torch.autograd.Function):
    """out[b,o,n] = max_k act(norm(Q[b,o,idx[b,n,k]] - Qc[b,o,n] + bias[o])) for the neighbour-dependent half of
    EdgeConv's grouped 1x1 convolution (csrc/gkg_edge.hip; reference torch_vertex.py:82-101 + torch_nn.py:57-69).
    ``bn``: a BatchNorm module or None; gamma / beta / running statistics are the slices of its tensors that belong to these
    O channels (``sl``).  Train-mode statistics run over all B*N*k elements, like the reference's BN on the (B,O,N,k) tensor."""

    @staticmethod
    def forward(ctx, qs, qc, nn_idx, bias, gamma, beta, bn, sl, act):
        lib = _lib.load()
        _need_cuda(qs, qc, nn_idx)
        B, O, M = qs.shape
        N, k = nn_idx.shape[1:]
        qs, qc, nn_idx = qs.contiguous(), qc.contiguous(), nn_idx.contiguous()
        dev = qs.device
        cnt = B * N * k
        bias_v = torch.zeros(O, dtype=torch.float32, device=dev) if bias is None else bias.detach().float()
        mean0 = invstd = None
        train_stats = bn is not None and (bn.training or not bn.track_running_stats)
        if train_stats:
            sums = torch.zeros(2 * O, dtype=torch.float64, device=dev)
            _lib.check(lib.gkg_edge_stats(_ptr(qs), _ptr(qc), _ptr(nn_idx), _ptr(sums), B, O, N, M, k, _stream()), "gkg_edge_stats")
            m0 = sums[:O] / cnt
            var = (sums[O:] / cnt - m0 * m0).clamp_min_(0.0)
            mean0, invstd = m0.float(), torch.rsqrt(var + bn.eps).float()
            a = gamma.detach().float() * invstd
            c = beta.detach().float() - a * mean0                      # the conv bias cancels against the batch mean
            if bn.training and bn.track_running_stats:
                mom = bn.momentum
                with torch.no_grad():
                    bn.running_mean[sl].mul_(1 - mom).add_(mom * (mean0 + bias_v))
                    bn.running_var[sl].mul_(1 - mom).add_(mom * (var * (cnt / max(cnt - 1, 1))).float())
        elif bn is not None:                                          # eval: fixed statistics
            invstd = torch.rsqrt(bn.running_var[sl].float() + bn.eps)
            mean0 = bn.running_mean[sl].float() - bias_v
            a = gamma.detach().float() * invstd
            c = beta.detach().float() - a * mean0
        else:
            a = torch.ones(O, dtype=torch.float32, device=dev)
            c = bias_v.clone()
        out = torch.empty((B, O, N), dtype=torch.float32, device=dev)
        need = any(ctx.needs_input_grad[:6])
        arg = torch.empty((B, O, N), dtype=torch.uint8, device=dev) if need else None
        _lib.check(lib.gkg_edge_fwd(_ptr(qs), _ptr(qc), _ptr(nn_idx), _ptr(a), _ptr(c), _ptr(out), _ptr(arg), B, O, N, M, k, act,
                                    _stream()), "gkg_edge_fwd")
        ctx.save_for_backward(qs, qc, nn_idx, arg, a, c, mean0, invstd)
        ctx.meta = (B, O, N, M, k, act, cnt, train_stats, bn is not None, bias is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        qs, qc, nn_idx, arg, a, c, mean0, invstd = ctx.saved_tensors
        B, O, N, M, k, act, cnt, train_stats, has_bn, has_bias = ctx.meta
        g = g.contiguous().float()
        dev = g.device
        dgamma = dbeta = dbias = None
        mg = mgz = None
        if has_bn:
            sums = torch.zeros(2 * O, dtype=torch.float64, device=dev)
            _lib.check(lib.gkg_edge_bwd_stats(_ptr(g), _ptr(qs), _ptr(qc), _ptr(nn_idx), _ptr(arg), _ptr(a), _ptr(c), _ptr(mean0),
                                              _ptr(invstd), _ptr(sums), B, O, N, M, k, act, _stream()), "gkg_edge_bwd_stats")
            dbeta, dgamma = sums[:O].float(), sums[O:].float()
            if train_stats:
                mg, mgz = (sums[:O] / cnt).float(), (sums[O:] / cnt).float()
            elif has_bias:
                dbias = a * dbeta                                     # eval: u = a (z + bias - running_mean) + beta
        dqs = torch.zeros_like(qs)
        dqc = torch.empty_like(qc)
        _lib.check(lib.gkg_edge_bwd(_ptr(g), _ptr(qs), _ptr(qc), _ptr(nn_idx), _ptr(arg), _ptr(a), _ptr(c), _ptr(mean0), _ptr(invstd),
                                    _ptr(mg), _ptr(mgz), _ptr(dqs), _ptr(dqc), B, O, N, M, k, act, _stream()), "gkg_edge_bwd")
        if not has_bn and has_bias:
            dbias = -dqc.sum(dim=(0, 2))                              # sum of the winning-edge gradients
        if has_bn and train_stats and has_bias:
            dbias = torch.zeros(O, dtype=torch.float32, device=dev)   # exactly zero: BN removes the mean
        return dqs, dqc, None, dbias, dgamma, dbeta, None, None, None


def edge_aggregate(qs, qc, nn_idx, bias, gamma, beta, bn, sl, act):
    return _EdgeAggregate.apply(qs, qc, nn_idx, bias, gamma, beta, bn, sl, act)
