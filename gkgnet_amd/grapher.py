"""Grapher / GrapherLabel: the Group-KNN graph-convolution blocks of GKGNet, drop-in for the reference's
modules of the same name (mmcls/models/backbones/vig_model/torch_vertex.py:278-403): same constructor
signatures, same state_dict keys, same outputs; k-NN + aggregation run on the HIP kernels."""
from __future__ import annotations

import torch
from torch import nn

from . import block, fused
from .dense import PointwiseConv2d
from .graph import DyGraphConv2d, DyGraphConv2dMultiGroup, DyGraphLabel, DyGraphLabelMultiGroup
from .layers import DropPath, act_layer, build_norm
from .relpos import build_relative_pos, resize_relative_pos


def _conv_norm(cin, cout):
    return nn.Sequential(PointwiseConv2d(cin, cout, 1, stride=1, padding=0), build_norm(cout))


class Grapher(nn.Module):
    """fc1 -> dynamic (grouped, dilated) k-NN graph conv -> fc2 -> + residual   (torch_vertex.py:278-333)."""

    def __init__(self, in_channels, kernel_size=9, dilation=1, conv="edge", act="relu", norm=None,
                 bias=True, stochastic=False, epsilon=0.0, r=1, n=196, drop_path=0.0,
                 relative_pos=False, use_multi_group=False, num_group=2):
        super().__init__()
        self.channels, self.n, self.r = in_channels, n, r
        self.fc1 = _conv_norm(in_channels, in_channels)
        if use_multi_group:
            self.graph_conv = DyGraphConv2dMultiGroup(in_channels, in_channels * 2, kernel_size, dilation, conv,
                                                      act, norm, bias, stochastic, epsilon, r, num_head=num_group)
        else:
            self.graph_conv = DyGraphConv2d(in_channels, in_channels * 2, kernel_size, dilation, conv,
                                            act, norm, bias, stochastic, epsilon, r)
        self.fc2 = _conv_norm(in_channels * 2, in_channels)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.relative_pos = None
        if relative_pos:
            self.relative_pos = nn.Parameter(build_relative_pos(in_channels, n, r), requires_grad=False)

    def _get_relative_pos(self, relative_pos, H, W):
        return resize_relative_pos(relative_pos, self.n, self.r, H, W)

    def forward(self, x):
        out = block.try_grapher(self, x)              # a step this module has taken before: straight to the block driver
        if out is not None:
            return out
        H, W = x.shape[2:]
        relative_pos = self._get_relative_pos(self.relative_pos, H, W)
        groups = self.graph_conv.num_head
        if fused.fused_supported(self, x, groups):
            return fused.grapher_forward(self, x, relative_pos, groups, want_edge=False)[0]
        shortcut = x
        x = self.fc1(x)
        x, _ = self.graph_conv(x, relative_pos)
        x = self.fc2(x)
        return self.drop_path(x) + shortcut


class FFNLabel(nn.Module):
    """1x1 conv MLP on label tokens; returns (B, L, C)  (torch_vertex.py:334-360)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act="relu", drop_path=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = _conv_norm(in_features, hidden_features)
        self.act = act_layer(act)
        self.fc2 = _conv_norm(hidden_features, out_features)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def forward(self, x, y=None):
        shortcut = x
        x = self.fc2(self.act(self.fc1(x)))
        x = self.drop_path(x) + shortcut
        return x.transpose(2, 1).squeeze(-1)


class GrapherLabel(nn.Module):
    """Label tokens attend to their k nearest image tokens (torch_vertex.py:361-403).
    forward(E (B,L,C), features (B,C,H,W)) -> (E' (B,L,C), edge_index)."""

    def __init__(self, in_channels, kernel_size=9, dilation=1, conv="edge", act="relu", norm=None,
                 bias=True, stochastic=False, epsilon=0.0, r=1, n=196, drop_path=0.0, relative_pos=False,
                 num_nodes=80, use_multi_group=False, num_group=2):
        super().__init__()
        self.channels = in_channels
        self.fc1 = _conv_norm(in_channels, in_channels)
        self.fc2 = _conv_norm(in_channels * 2, in_channels)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        if not use_multi_group:
            self.graph_conv = DyGraphLabel(in_channels, in_channels * 2, kernel_size, dilation, conv,
                                           act, norm, bias, stochastic, epsilon, r)
        else:
            self.graph_conv = DyGraphLabelMultiGroup(in_channels, in_channels * 2, kernel_size, dilation, conv,
                                                     act, norm, bias, stochastic, epsilon, r, num_head=num_group)
        self.ffn = FFNLabel(in_channels, in_channels * 4, act=act, drop_path=drop_path)

    def forward(self, x, features):
        res = block.try_label(self, x, features)      # (see Grapher.forward)
        if res is not None:
            from .graph import DyGraphLabel
            return res[0], (res[1] if isinstance(self.graph_conv, DyGraphLabel) else res[1][0])
        B, C = features.shape[:2]
        groups = self.graph_conv.num_head
        if fused.fused_supported(self, x, groups) and features.is_cuda and features.dtype in (torch.float32, torch.bfloat16, torch.float16):
            out, edge = fused.grapher_label_forward(self, x, features, groups)
            from .graph import DyGraphLabel
            return out, (edge if isinstance(self.graph_conv, DyGraphLabel) else edge[0])
        features = features.reshape(B, C, -1)
        x = x.transpose(2, 1).unsqueeze(-1)
        shortcut = x
        x = self.fc1(x)
        x, edge_index = self.graph_conv(x, features)
        x = self.fc2(x)
        x = self.drop_path(x) + shortcut
        return self.ffn(x), edge_index
