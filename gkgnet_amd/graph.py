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

    def _hip_plan(self, x):
        """(conv, bn or None, act code) when the HIP aggregation applies: fp32 CUDA tensors, BasicConv = grouped 1x1 conv
        [+ BatchNorm with a fixed momentum, statistics local to this process] [+ GELU / ReLU], no dropout."""
        from . import ops
        if not (x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled()):
            return None
        mods = list(self.nn)
        conv = mods[0]
        if not isinstance(conv, nn.Conv2d) or conv.groups != 4 or conv.out_channels % 4 or conv.in_channels % 4:
            return None
        bn, act = None, ops.ACT_NONE
        for m in mods[1:]:
            if isinstance(m, nn.modules.batchnorm._BatchNorm):
                if m.momentum is None or not m.affine or (m.training and torch.distributed.is_available()
                                                          and torch.distributed.is_initialized()
                                                          and torch.distributed.get_world_size() > 1
                                                          and isinstance(m, nn.SyncBatchNorm)):
                    return None
                bn = m
            elif isinstance(m, nn.GELU) and getattr(m, "approximate", "none") == "none":
                act = ops.ACT_GELU
            elif isinstance(m, nn.ReLU):
                act = ops.ACT_RELU
            else:
                return None
        return conv, bn, act

    def forward(self, x, edge_index, y=None):
        bg, c = x.shape[:2]
        xt = x.reshape(bg, c, -1)
        src = xt if y is None else y.reshape(bg, c, -1)
        idx = edge_index[0]
        n, k = idx.shape[1:]
        plan = self._hip_plan(x)
        if plan is not None and k <= 255:
            return self._forward_hip(xt, src, idx, *plan)
        x_j = torch.gather(src, 2, idx.reshape(bg, 1, n * k).expand(bg, c, n * k)).reshape(bg, c, n, k)
        x_i = xt.unsqueeze(-1).expand(-1, -1, -1, k)
        return self.nn(torch.cat([x_i, x_j - x_i], dim=1)).max(dim=-1, keepdim=True).values

    def _forward_hip(self, xt, src, idx, conv, bn, act):
        """The same function without the (B, 2C, N, k) tensor.  Groups 0/1 of the grouped convolution see only x_i: a per-node
        projection, constant over k, so the max is the value itself and its BN statistics over (B, N, k) equal those over
        (B, N).  Groups 2/3 see only x_j - x_i: z = Q[idx] - Qc + bias with Q = W src, Qc = W x — two per-node projections
        and the gather kernels of csrc/gkg_edge.hip (gkg_edge_*)."""
        from . import ops
        B, C, N = xt.shape
        k = idx.shape[2]
        O = conv.out_channels
        Oh, ci = O // 2, conv.in_channels // 4                  # channels of each half; inputs per group (= C / 2)
        W = conv.weight.view(O, ci)
        bias = conv.bias
        # groups 0 and 1: inputs x[:, :C/2] and x[:, C/2:]
        zA = torch.cat([torch.einsum("oc,bcn->bon", W[:O // 4], xt[:, :ci]),
                        torch.einsum("oc,bcn->bon", W[O // 4:Oh], xt[:, ci:])], dim=1)
        if bias is not None:
            zA = zA + bias[:Oh].view(1, -1, 1)
        cnt = B * N * k
        if bn is not None:
            if bn.training or not bn.track_running_stats:
                mean = zA.mean(dim=(0, 2))
                var = zA.var(dim=(0, 2), unbiased=False)
                if bn.training and bn.track_running_stats:
                    with torch.no_grad():
                        bn.running_mean[:Oh].mul_(1 - bn.momentum).add_(bn.momentum * mean)
                        bn.running_var[:Oh].mul_(1 - bn.momentum).add_(bn.momentum * var * (cnt / max(cnt - 1, 1)))
                        bn.num_batches_tracked += 1
            else:
                mean, var = bn.running_mean[:Oh], bn.running_var[:Oh]
            zA = (zA - mean.view(1, -1, 1)) * torch.rsqrt(var + bn.eps).view(1, -1, 1) * bn.weight[:Oh].view(1, -1, 1) \
                + bn.bias[:Oh].view(1, -1, 1)
        if act == ops.ACT_GELU:
            zA = torch.nn.functional.gelu(zA)
        elif act == ops.ACT_RELU:
            zA = torch.relu(zA)
        # groups 2 and 3: per-node projections of the source / centre tokens, then the gather kernels
        def proj(t):
            return torch.cat([torch.einsum("oc,bcn->bon", W[Oh:Oh + O // 4], t[:, :ci]),
                              torch.einsum("oc,bcn->bon", W[Oh + O // 4:], t[:, ci:])], dim=1)
        qs = proj(src)
        qc = qs if src is xt else proj(xt)
        sl = slice(Oh, O)
        zB = ops.edge_aggregate(qs, qc, idx, None if bias is None else bias[sl],
                                None if bn is None else bn.weight[sl], None if bn is None else bn.bias[sl], bn, sl, act)
        return torch.cat([zA, zB], dim=1).unsqueeze(-1)


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
