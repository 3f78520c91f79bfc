"""TEST INFRASTRUCTURE — CPU oracle (PyTorch fp32) for the Group-KNN graph-convolution path.

This file is a *functional restatement* of the reference's algorithm for the hot
path, written against the reference's op order so that fp32 rounding matches:
every function cites the reference file:line it follows (paths relative to the
reference tree, jin-s13/GKGNet @ 2024-10-22).  It is pinned against golden
vectors that the reference itself produced in the build container
(``tools/gen_golden.py`` -> ``tests/golden/*.npz``; see tests/test_oracle_golden.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product (``gkgnet_amd``) never does.

Layout conventions (same as the reference):
  x        (BG, c, N, 1) or (BG, c, N)   query tokens, channel-major
  y        (BG, c, M, 1) or None         key tokens (None -> self graph, M = N)
  relpos   (1, N, M) fp32 or None
  nn_idx   (BG, N, k) int64              neighbour (key) index per query, ascending distance
Parameters are passed as a flat ``dict`` with the reference's ``state_dict`` keys.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- k-NN
def l2_normalize_tokens(t: Tensor) -> Tensor:
    """F.normalize(t, p=2, dim=1), eps 1e-12 — torch_edge.py:167-168,173."""
    return F.normalize(t, p=2.0, dim=1)


def pairwise_sqdist(xq: Tensor, yk: Tensor) -> Tensor:
    """dist[b,n,m] = |x_n|^2 - 2 x_n.y_m + |y_m|^2 in the reference's op order.

    xq (BG,N,c), yk (BG,M,c).  Order ``(x_sq + inner) + y_sq^T`` follows
    torch_edge.py:17-20 (self) / :47-51 (xy).
    """
    inner = -2 * torch.matmul(xq, yk.transpose(2, 1))
    x_sq = torch.sum(torch.mul(xq, xq), dim=-1, keepdim=True)
    y_sq = torch.sum(torch.mul(yk, yk), dim=-1, keepdim=True)
    return x_sq + inner + y_sq.transpose(2, 1)


def knn_distances(x: Tensor, y: Optional[Tensor], relpos: Optional[Tensor],
                  normalize: bool = True) -> Tensor:
    """Full (BG,N,M) distance matrix the reference feeds to topk.

    torch_edge.py:164-176 (normalise), :54-86 / :89-106 (distance + ``+= relative_pos``).
    The reference's N>10000 query chunking (:66-78) is a memory device and does not
    change any value, so it is not restated.
    """
    x = x.reshape(x.shape[0], x.shape[1], -1, 1)
    if y is not None:
        y = y.reshape(y.shape[0], y.shape[1], -1, 1)
    with torch.no_grad():
        if normalize:
            x = l2_normalize_tokens(x)
            if y is not None:
                y = l2_normalize_tokens(y)
        xq = x.transpose(2, 1).squeeze(-1)
        yk = xq if y is None else y.transpose(2, 1).squeeze(-1)
        dist = pairwise_sqdist(xq, yk)
        if relpos is not None:
            dist = dist + relpos
    return dist


def knn_graph(x: Tensor, y: Optional[Tensor], relpos: Optional[Tensor], k: int,
              dilation: int = 1, normalize: bool = True) -> Tensor:
    """edge_index (2, BG, N, k): [0] neighbour idx (every ``dilation``-th of the sorted
    top k*dilation), [1] centre idx.  torch_edge.py:83-86,104-106 + DenseDilated :146-148."""
    dist = knn_distances(x, y, relpos, normalize)
    bg, n, _ = dist.shape
    _, nn_idx = torch.topk(-dist, k=k * dilation)
    center = torch.arange(n, device=dist.device).view(1, n, 1).expand(bg, n, k * dilation)
    edge = torch.stack((nn_idx, center), dim=0)
    return edge[:, :, :, ::dilation]


# ------------------------------------------------------------------ max-relative
def gather_tokens(src: Tensor, idx: Tensor) -> Tensor:
    """out[b,ch,n,j] = src[b,ch,idx[b,n,j]] — torch_nn.py:84-105 (same result, no flatten)."""
    bg, c = src.shape[:2]
    s = src.reshape(bg, c, -1)
    n, k = idx.shape[1:]
    flat = idx.reshape(bg, 1, n * k).expand(bg, c, n * k)
    return torch.gather(s, 2, flat).reshape(bg, c, n, k)


def max_relative(x: Tensor, nn_idx: Tensor, y: Optional[Tensor]) -> Tensor:
    """m[b,ch,n] = max_k (src[b,ch,idx[b,n,k]] - x[b,ch,n]) — torch_vertex.py:49-54."""
    bg, c = x.shape[:2]
    xc = x.reshape(bg, c, -1)
    src = xc if y is None else y.reshape(bg, c, -1)
    x_j = gather_tokens(src, nn_idx)
    rel = x_j - xc.unsqueeze(-1)
    return rel.max(dim=-1).values


def interleave_channels(x: Tensor, m: Tensor, full_c: int) -> Tensor:
    """(BG,c,N)+(BG,c,N) -> (B, 2C, N, 1) as [x0,m0,x1,m1,...] — torch_vertex.py:57-61."""
    n = x.shape[-1] if x.dim() == 3 else x.shape[2]
    xb = x.reshape(-1, full_c, n)
    mb = m.reshape(-1, full_c, n)
    return torch.stack((xb, mb), dim=2).reshape(xb.shape[0], 2 * full_c, n, 1)


# ------------------------------------------------------------------ dense pieces
class _DenseGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        return g.contiguous()


def _bn(t: Tensor, p: Dict[str, Tensor], prefix: str, training: bool) -> Tensor:
    """BatchNorm2d / SyncBatchNorm without a process group (eps 1e-5, momentum 0.1).
    Running statistics are NOT updated here (oracle is stateless).
    Input and incoming gradient are made dense NCHW first: torch 2.10's CPU batch_norm backward is wrong
    for mixed memory formats (see tools/gen_golden.py); values and the forward result are unaffected."""
    out = F.batch_norm(t.contiguous(), p[prefix + ".running_mean"].clone(), p[prefix + ".running_var"].clone(),
                       p[prefix + ".weight"], p[prefix + ".bias"], training, 0.1, 1e-5)
    return _DenseGrad.apply(out) if out.requires_grad else out


def conv_bn(t: Tensor, p: Dict[str, Tensor], prefix: str, training: bool, groups: int = 1) -> Tensor:
    """Sequential(Conv2d 1x1, norm) — torch_vertex.py:290-294,302-306."""
    t = F.conv2d(t, p[prefix + ".0.weight"], p.get(prefix + ".0.bias"), groups=groups)
    return _bn(t, p, prefix + ".1", training)


def basic_conv(t: Tensor, p: Dict[str, Tensor], prefix: str, training: bool) -> Tensor:
    """BasicConv([2C,2C], 'gelu', 'batch'): Conv2d(1x1, groups=4)+BN+GELU(erf) — torch_nn.py:57-69."""
    return F.gelu(conv_bn(t, p, prefix, training, groups=4))


def mr_conv(x: Tensor, edge_index: Tensor, y: Optional[Tensor], p: Dict[str, Tensor], prefix: str,
            full_c: int, training: bool) -> Tuple[Tensor, Tensor]:
    """MRConv2d.forward — torch_vertex.py:47-62.  Returns (BasicConv out (B,2C,N,1), m (BG,c,N))."""
    bg, c = x.shape[:2]
    m = max_relative(x, edge_index[0], y)
    cat = interleave_channels(x.reshape(bg, c, -1), m, full_c)
    return basic_conv(cat, p, prefix + ".nn", training), m


def edge_conv(x: Tensor, edge_index: Tensor, y: Optional[Tensor], p: Dict[str, Tensor], prefix: str,
              training: bool) -> Tensor:
    """EdgeConv2d.forward — torch_vertex.py:91-101 (G=1 only)."""
    bg, c = x.shape[:2]
    xc = x.reshape(bg, c, -1)
    src = xc if y is None else y.reshape(bg, c, -1)
    x_i = xc.unsqueeze(-1).expand(-1, -1, -1, edge_index.shape[-1])
    x_j = gather_tokens(src, edge_index[0])
    h = basic_conv(torch.cat([x_i, x_j - x_i], dim=1), p, prefix + ".nn", training)
    return h.max(dim=-1, keepdim=True).values


# ----------------------------------------------------------------- Grapher family
def dy_graph_conv(x: Tensor, relpos: Optional[Tensor], p: Dict[str, Tensor], prefix: str, *,
                  k: int, dilation: int, r: int, groups: int, conv: str, training: bool):
    """DyGraphConv2d[MultiGroup].forward — torch_vertex.py:191-205 / 218-228."""
    b, c, h, w = x.shape
    y = None
    if r > 1:
        y = F.avg_pool2d(x, r, r).reshape(b, c, -1, 1)
    xg = x.reshape(b * groups, c // groups, -1, 1)
    yg = None if y is None else y.reshape(b * groups, c // groups, -1, 1)
    edge_index = knn_graph(xg, yg, relpos, k, dilation)
    if conv == "mr":
        out, m = mr_conv(xg, edge_index, yg, p, prefix + ".gconv", c, training)
    elif conv == "edge":
        assert groups == 1
        out, m = edge_conv(xg, edge_index, yg, p, prefix + ".gconv", training), None
    else:
        raise NotImplementedError(conv)
    return out.reshape(b, -1, h, w), edge_index, m


def grapher_forward(x: Tensor, p: Dict[str, Tensor], *, k: int = 9, dilation: int = 1, r: int = 1,
                    groups: int = 1, conv: str = "mr", training: bool = True,
                    return_aux: bool = False):
    """Grapher.forward — torch_vertex.py:325-333 (drop_path = 0).  ``groups`` = num_group when
    use_multi_group else 1.  ``p['relative_pos']`` (1,N,M) optional."""
    relpos = p.get("relative_pos")
    h1 = conv_bn(x, p, "fc1", training)
    g, edge_index, m = dy_graph_conv(h1, relpos, p, "graph_conv", k=k, dilation=dilation, r=r,
                                     groups=groups, conv=conv, training=training)
    out = conv_bn(g, p, "fc2", training) + x
    if return_aux:
        return out, dict(fc1=h1, edge_index=edge_index, m=m, graph=g)
    return out


def ffn_label_forward(x: Tensor, p: Dict[str, Tensor], prefix: str, training: bool) -> Tensor:
    """FFNLabel.forward — torch_vertex.py:352-360 (act = gelu, drop_path = 0)."""
    h = F.gelu(conv_bn(x, p, prefix + ".fc1", training))
    h = conv_bn(h, p, prefix + ".fc2", training) + x
    return h.transpose(2, 1).squeeze(-1)


def grapher_label_forward(e: Tensor, feat: Tensor, p: Dict[str, Tensor], *, k: int = 9,
                          groups: int = 1, use_multi_group: bool = True, training: bool = True,
                          return_aux: bool = False):
    """GrapherLabel.forward — torch_vertex.py:392-403 with DyGraphLabel[MultiGroup] :243-251/:266-275.
    e (B,L,C) label tokens, feat (B,C,H,W).  Returns (e' (B,L,C), graph) where graph is nn_idx
    (BG,L,k) for the multi-group module (:275 returns edge_index[0]) and the full (2,B,L,k)
    edge_index for the single-group DyGraphLabel (:251 returns edge_index) — a reference quirk."""
    b, c = feat.shape[:2]
    y = feat.reshape(b, c, -1)
    x0 = e.transpose(2, 1).unsqueeze(-1)
    h1 = conv_bn(x0, p, "fc1", training)
    xg = h1.reshape(b * groups, c // groups, -1, 1)
    yg = y.reshape(b * groups, c // groups, -1, 1)
    edge_index = knn_graph(xg, yg, None, k, 1)
    g, m = mr_conv(xg, edge_index, yg, p, "graph_conv.gconv", c, training)
    g = g.reshape(b, 2 * c, -1, 1)
    h2 = conv_bn(g, p, "fc2", training) + x0
    out = ffn_label_forward(h2, p, "ffn", training)
    graph = edge_index[0] if use_multi_group else edge_index
    if return_aux:
        return out, graph, dict(fc1=h1, m=m, graph=g)
    return out, graph


# ------------------------------------------------------------ relative_pos constant
def sincos_1d(dim: int, pos: np.ndarray) -> np.ndarray:
    """pos_embed.py:66-85: [sin(pos*w) | cos(pos*w)], w_i = 10000^(-i/(dim/2)), float64."""
    omega = np.arange(dim // 2, dtype=np.float64)
    omega /= dim / 2.0
    omega = 1.0 / 10000 ** omega
    out = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(out), np.cos(out)], axis=1)


def relative_pos_matrix(embed_dim: int, grid: int) -> np.ndarray:
    """get_2d_relative_pos_embed — pos_embed.py:21-29,38-63: 2*PE*PE^T/D, 'w first' meshgrid."""
    gh = np.arange(grid, dtype=np.float32)
    gw = np.arange(grid, dtype=np.float32)
    mesh = np.stack(np.meshgrid(gw, gh), axis=0).reshape(2, 1, grid, grid)
    pe = np.concatenate([sincos_1d(embed_dim // 2, mesh[0]), sincos_1d(embed_dim // 2, mesh[1])], axis=1)
    return 2 * np.matmul(pe, pe.transpose()) / pe.shape[1]


def grapher_relative_pos(in_channels: int, n: int, r: int) -> Tensor:
    """The frozen ``relative_pos`` Parameter (1, n, n/r^2) — torch_vertex.py:309-315:
    bicubic resize of the (n,n) matrix *as an image*, then negated."""
    rp = torch.from_numpy(np.float32(relative_pos_matrix(in_channels, int(n ** 0.5)))).unsqueeze(0).unsqueeze(1)
    rp = F.interpolate(rp, size=(n, n // (r * r)), mode="bicubic", align_corners=False)
    return -rp.squeeze(1)


def runtime_relative_pos(relpos: Optional[Tensor], n_built: int, r: int, h: int, w: int):
    """Grapher._get_relative_pos — torch_vertex.py:317-323."""
    if relpos is None or h * w == n_built:
        return relpos
    n = h * w
    return F.interpolate(relpos.unsqueeze(0), size=(n, n // (r * r)), mode="bicubic").squeeze(0)
