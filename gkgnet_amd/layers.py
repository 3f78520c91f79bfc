"""Small dense building blocks with the reference's constructor signatures and state_dict keys
(reference mmcls/models/backbones/vig_model/torch_nn.py:13-81).  Dense 1x1 projections run on the
MFMA units through torch's ROCm GEMM/conv libraries."""
from __future__ import annotations

import torch
from torch import nn

from .dense import PointwiseConv2d

# Reference default (torch_nn.py:8, torch_vertex.py:14, gkgnet.py:23).  Without an initialised process
# group SyncBatchNorm behaves as plain BatchNorm; set type='BN' to force local statistics under DDP.
norm_cfg = dict(type="SyncBN", requires_grad=True)


class _ContiguousGrad(torch.autograd.Function):
    """Identity whose backward hands a dense NCHW gradient upstream."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g if g.is_contiguous() else g.contiguous()


def _guard_memory_format(norm_forward):
    """PyTorch-ROCm 2.10 computes a WRONG batch-norm backward when the input and the incoming gradient
    have different memory formats (one NCHW-contiguous, the other a channels-last strided view — exactly
    what the label path's (B,L,C)<->(B,C,L,1) transposes produce; measured on MI355X: dx off by O(10)).
    Make both dense NCHW around every norm layer.  No-ops (no copies) when tensors already are."""

    def forward(self, x):
        if not x.is_contiguous():
            x = x.contiguous()
        y = norm_forward(self, x)
        if torch.is_grad_enabled() and y.requires_grad:
            y = _ContiguousGrad.apply(y)
        return y

    return forward


class _StatsEpoch:
    """``_gkg_epoch`` changes whenever the running statistics may have changed behind PyTorch's version counters: the fused
    path's kernels update them through raw pointers (and a captured training step does so without running any Python), so
    every train() / eval() switch and every eager train-mode update bumps it.  Inference-time caches derived from the
    statistics (fused._bn_eval_ac, fused._folded_of) are keyed on it."""
    _gkg_epoch = 0

    def train(self, mode: bool = True):
        self._gkg_epoch += 1
        return super().train(mode)


class GuardedSyncBatchNorm(_StatsEpoch, nn.SyncBatchNorm):
    forward = _guard_memory_format(nn.SyncBatchNorm.forward)


class GuardedBatchNorm2d(_StatsEpoch, nn.BatchNorm2d):
    forward = _guard_memory_format(nn.BatchNorm2d.forward)


def build_norm(num_features: int) -> nn.Module:
    kind = norm_cfg.get("type", "SyncBN")
    if kind == "SyncBN":
        layer = GuardedSyncBatchNorm(num_features)
    elif kind == "BN":
        layer = GuardedBatchNorm2d(num_features)
    else:
        raise NotImplementedError(f"norm type {kind}")
    for p in layer.parameters():
        p.requires_grad = bool(norm_cfg.get("requires_grad", True))
    return layer


def act_layer(act: str, inplace: bool = False, neg_slope: float = 0.2, n_prelu: int = 1) -> nn.Module:
    act = act.lower()
    table = {
        "relu": lambda: nn.ReLU(inplace),
        "leakyrelu": lambda: nn.LeakyReLU(neg_slope, inplace),
        "prelu": lambda: nn.PReLU(num_parameters=n_prelu, init=neg_slope),
        "gelu": lambda: nn.GELU(),
        "hswish": lambda: nn.Hardswish(inplace),
    }
    if act not in table:
        raise NotImplementedError("activation layer [%s] is not found" % act)
    return table[act]()


def norm_layer(norm: str, nc: int) -> nn.Module:
    norm = norm.lower()
    if norm == "batch":
        return build_norm(nc)
    if norm == "instance":
        return nn.InstanceNorm2d(nc, affine=False)
    raise NotImplementedError("normalization layer [%s] is not found" % norm)


class BasicConv(nn.Sequential):
    """[Conv2d(1x1, groups=4) -> norm -> act -> Dropout2d] per consecutive channel pair.
    ``groups=4`` is fixed and independent of the k-NN group count, as in the reference (torch_nn.py:61)."""

    def __init__(self, channels, act="relu", norm=None, bias=True, drop=0.0):
        layers = []
        for cin, cout in zip(channels[:-1], channels[1:]):
            layers.append(PointwiseConv2d(cin, cout, 1, bias=bias, groups=4))
            if norm is not None and norm.lower() != "none":
                layers.append(norm_layer(norm, channels[-1]))
            if act is not None and act.lower() != "none":
                layers.append(act_layer(act))
            if drop > 0:
                layers.append(nn.Dropout2d(drop))
        super().__init__(*layers)
        self.reset_parameters()

    def reset_parameters(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, (nn.modules.batchnorm._BatchNorm, nn.InstanceNorm2d)):
                if m.weight is not None:
                    nn.init.ones_(m.weight)
                    nn.init.zeros_(m.bias)


class DropPath(nn.Module):
    """Per-sample stochastic depth (identity at p == 0 or in eval mode)."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def active(self) -> bool:
        return self.drop_prob > 0.0 and self.training

    def sample_scale(self, batch: int, device, dtype=torch.float32):
        """Per-image factor mask / keep (mask ~ Bernoulli(keep)), shape (batch,); None when inactive.  The fused block
        folds it into its last BN-apply kernel; ``forward`` draws from the same place, so both paths consume the RNG alike."""
        if not self.active():
            return None
        keep = 1.0 - self.drop_prob
        return torch.empty(batch, device=device, dtype=dtype).bernoulli_(keep) / keep

    def forward(self, x):
        s = self.sample_scale(x.shape[0], x.device, x.dtype)
        if s is None:
            return x
        return x * s.view((x.shape[0],) + (1,) * (x.dim() - 1))

    def extra_repr(self):
        return f"p={self.drop_prob}"
