"""Graph-convolution modules of the Group-KNN path with the reference's class names, constructor
signatures and state_dict keys (reference mmcls/models/backbones/vig_model/torch_edge.py:126-176 and
torch_vertex.py:38-275), but with the k-NN search and the neighbour aggregation executed by the
HIP kernels in libgkg_hip.so (gkgnet_amd.ops) instead of ATen matmul/topk/index/max chains."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .layers import BasicConv


class DenseDilatedKnnGraph(nn.Module):
    """edge_index = dilated k-NN of L2-normalised tokens (reference torch_edge.py:152-176).
    ``stochastic`` keeps the reference's train-time random neighbour subsampling (torch_edge.py:139-145)."""

    def __init__(self, k=9, dilation=1, stochastic=False, epsilon=0.0):
        super().__init__()
        self.k, self.dilation, self.stochastic, self.epsilon = k, dilation, stochastic, epsilon

    def forward(self, x, y=None, relative_pos=None):
        if self.stochastic and self.training and torch.rand(1).item() < self.epsilon:
            full = ops.knn_graph(x, y, relative_pos, self.k * self.dilation, 1)
            pick = torch.randperm(self.k * self.dilation, device=full.device)[: self.k]
            return full[:, :, :, pick]
        return ops.knn_graph(x, y, relative_pos, self.k, self.dilation)


def _interleave(x, m, full_c):
    """(BG,c,N) x2 -> (B, 2C, N, 1) with channels [x0, m0, x1, m1, ...] (reference torch_vertex.py:57-61)."""
    n = x.shape[-1]
    xb = x.reshape(-1, full_c, n)
    mb = m.reshape(-1, full_c, n)
    return torch.stack((xb, mb), dim=2).reshape(xb.shape[0], 2 * full_c, n, 1)


class MRConv2d(nn.Module):
    """Max-relative graph convolution (reference torch_vertex.py:38-62)."""

    def __init__(self, in_channels, out_channels, act="relu", norm=None, bias=True):
        super().__init__()
        self.in_channels = in_channels
        self.nn = BasicConv([in_channels * 2, out_channels], act, norm, bias)

    def forward(self, x, edge_index, y=None):
        bg, c = x.shape[:2]
        xt = x.reshape(bg, c, -1)
        m = ops.max_relative(xt, edge_index[0], None if y is None else y.reshape(bg, c, -1))
        return self.nn(_interleave(xt, m, self.in_channels))


class EdgeConv2d(nn.Module):
    """EdgeConv: max_k nn([x_i, x_j - x_i]) (reference torch_vertex.py:82-101).  Secondary aggregation —
    GKGNet always selects 'mr'; only valid with a single group, like the reference."""

    def __init__(self, in_channels, out_channels, act="relu", norm=None, bias=True):
        super().__init__()
        self.in_channels = in_channels
        self.nn = BasicConv([in_channels * 2, out_channels], act, norm, bias)

    def forward(self, x, edge_index, y=None):
        bg, c = x.shape[:2]
        xt = x.reshape(bg, c, -1)
        src = xt if y is None else y.reshape(bg, c, -1)
        idx = edge_index[0]
        n, k = idx.shape[1:]
        x_j = torch.gather(src, 2, idx.reshape(bg, 1, n * k).expand(bg, c, n * k)).reshape(bg, c, n, k)
        x_i = xt.unsqueeze(-1).expand(-1, -1, -1, k)
        return self.nn(torch.cat([x_i, x_j - x_i], dim=1)).max(dim=-1, keepdim=True).values


class GraphConv2d(nn.Module):
    """Static graph convolution dispatcher (reference torch_vertex.py:153-173)."""

    def __init__(self, in_channels, out_channels, conv="edge", act="relu", norm=None, bias=True):
        super().__init__()
        if conv == "edge":
            self.gconv = EdgeConv2d(in_channels, out_channels, act, norm, bias)
        elif conv == "mr":
            self.gconv = MRConv2d(in_channels, out_channels, act, norm, bias)
        else:
            raise NotImplementedError("conv:{} is not supported".format(conv))

    def forward(self, x, edge_index, y=None):
        return self.gconv(x, edge_index, y)


class DyGraphConv2dMultiGroup(GraphConv2d):
    """Dynamic graph conv with G independent channel groups for the k-NN (reference torch_vertex.py:175-205)."""

    def __init__(self, in_channels, out_channels, kernel_size=9, dilation=1, conv="edge", act="relu",
                 norm=None, bias=True, stochastic=False, epsilon=0.0, r=1, num_head=2):
        super().__init__(in_channels, out_channels, conv, act, norm, bias)
        self.k, self.d, self.r, self.num_head = kernel_size, dilation, r, num_head
        self.dilated_knn_graph = DenseDilatedKnnGraph(kernel_size, dilation, stochastic, epsilon)

    def forward(self, x, relative_pos=None):
        B, C, H, W = x.shape
        g = self.num_head
        y = None
        if self.r > 1:
            y = F.avg_pool2d(x, self.r, self.r).reshape(B * g, C // g, -1, 1)
        x = x.reshape(B * g, C // g, -1, 1)
        edge_index = self.dilated_knn_graph(x, y, relative_pos)
        out = super().forward(x, edge_index, y)
        return out.reshape(B, -1, H, W), edge_index


class DyGraphConv2d(DyGraphConv2dMultiGroup):
    """Single-group variant (reference torch_vertex.py:206-228)."""

    def __init__(self, in_channels, out_channels, kernel_size=9, dilation=1, conv="edge", act="relu",
                 norm=None, bias=True, stochastic=False, epsilon=0.0, r=1):
        super().__init__(in_channels, out_channels, kernel_size, dilation, conv, act, norm, bias, stochastic,
                         epsilon, r, num_head=1)


class DyGraphLabelMultiGroup(GraphConv2d):
    """Label tokens (queries) -> image tokens (keys), no positional bias (reference torch_vertex.py:253-275).
    Returns (x (B,2C,L,1), nn_idx (B*G,L,k))."""

    def __init__(self, in_channels, out_channels, kernel_size=9, dilation=1, conv="edge", act="relu",
                 norm=None, bias=True, stochastic=False, epsilon=0.0, r=1, num_head=2, bit_graph=True):
        super().__init__(in_channels, out_channels, conv, act, norm, bias)
        self.k, self.d, self.r, self.num_head = kernel_size, dilation, r, num_head
        self.out_channels = out_channels
        self.dilated_knn_graph = DenseDilatedKnnGraph(kernel_size, dilation, stochastic, epsilon)

    def _run(self, x, y):
        B, C = x.shape[:2]
        g = self.num_head
        if y is not None:
            y = y.reshape(B * g, C // g, -1, 1)
        x = x.reshape(B * g, C // g, -1, 1)
        edge_index = self.dilated_knn_graph(x, y)
        out = super().forward(x, edge_index, y)
        return out.reshape(B, self.out_channels, -1, 1), edge_index

    def forward(self, x, y=None):
        out, edge_index = self._run(x, y)
        return out, edge_index[0]


class DyGraphLabel(DyGraphLabelMultiGroup):
    """Single-group variant; returns the full (2,B,L,k) edge_index like the reference (torch_vertex.py:229-251)."""

    def __init__(self, in_channels, out_channels, kernel_size=9, dilation=1, conv="edge", act="relu",
                 norm=None, bias=True, stochastic=False, epsilon=0.0, r=1):
        super().__init__(in_channels, out_channels, kernel_size, dilation, conv, act, norm, bias, stochastic,
                         epsilon, r, num_head=1)

    def forward(self, x, y):
        return self._run(x, y)
