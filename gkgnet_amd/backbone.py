"""GKGNet backbone (Pyramid-ViG + label-token branch) wired on the MI355X Grapher / GrapherLabel blocks.

Drop-in for the reference's ``@BACKBONES.register_module() class GKGNet`` (mmcls/models/backbones/gkgnet.py:120-284):
same constructor arguments, same module tree (=> same ``state_dict`` keys, so ``pvig_s_82.1.pth.tar`` and the
GKGNet-576 checkpoints load by key), same forward triple ``(label tokens (B,n_classes,C4), gap (B,C4),
edge_index)``.  ``arch_settings`` additionally has ``'m'`` (Pyramid-ViG-M, blocks [2,2,16,2], channels
[96,192,384,768]) which BASELINE config 5 asks for and the reference lacks.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import fused
from .grapher import Grapher, GrapherLabel
from .layers import DropPath, act_layer, build_norm
from .registry import BACKBONES, register_with_mmcls


class FFN(nn.Module):
    """1x1 conv MLP with BN, GELU and residual (reference gkgnet.py:46-72)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act="relu", drop_path=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Sequential(nn.Conv2d(in_features, hidden_features, 1, stride=1, padding=0), build_norm(hidden_features))
        self.act = act_layer(act)
        self.fc2 = nn.Sequential(nn.Conv2d(hidden_features, out_features, 1, stride=1, padding=0), build_norm(out_features))
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()

    def forward(self, x):
        if fused.ffn_supported(self, x):
            return fused.ffn_forward(self, x)
        return self.drop_path(self.fc2(self.act(self.fc1(x)))) + x


class Stem(nn.Module):
    """Image -> visual tokens: three 3x3 convs (strides 2, 2, 1) with BN and activation (reference gkgnet.py:74-100)."""

    def __init__(self, img_size=224, in_dim=3, out_dim=768, act="relu"):
        super().__init__()
        self.convs = nn.Sequential(
            nn.Conv2d(in_dim, out_dim // 2, 3, stride=2, padding=1), build_norm(out_dim // 2), act_layer(act),
            nn.Conv2d(out_dim // 2, out_dim, 3, stride=2, padding=1), build_norm(out_dim), act_layer(act),
            nn.Conv2d(out_dim, out_dim, 3, stride=1, padding=1), build_norm(out_dim))

    def forward(self, x):
        mods = list(self.convs)
        conv0, bn0 = mods[0], mods[1]
        act0 = mods[2] if not isinstance(mods[2], nn.Conv2d) else None
        first = None
        if fused.stem_conv_supported(conv0, x) and (act0 is None or isinstance(act0, nn.GELU)):
            # the first convolution (3 input channels) as a direct kernel: csrc/gkg_stem.hip
            ac16 = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
            if (not torch.is_grad_enabled() and not bn0.training and bn0.track_running_stats and fused._bn_ok(bn0)
                    and (ac16 or not torch.is_autocast_enabled())):
                first = fused.stem_conv_bn_act_eval(conv0, bn0, act0, x, ac16)           # conv + BN(eval) + GELU: one launch
                start = 3 if act0 is not None else 2
            elif not torch.is_autocast_enabled():
                first = fused.stem_conv(conv0, x, bn0)                                        # training: BN on the kernels below
                start = 1
        # inference: every remaining conv -> BN (-> GELU) unit is the library convolution + ONE own pass (fused.conv_bn_act_eval)
        if not torch.is_grad_enabled():
            units, i, ok = [], (start if first is not None else 0), True
            while i < len(mods) and ok:
                conv = mods[i]
                bn = mods[i + 1] if i + 1 < len(mods) else None
                act = mods[i + 2] if i + 2 < len(mods) and not isinstance(mods[i + 2], nn.Conv2d) else None
                ok = isinstance(conv, nn.Conv2d) and bn is not None and not isinstance(bn, nn.Conv2d)
                units.append((conv, bn, act))
                i += 2 if act is None else 3
            cur = first if first is not None else x
            if ok and units and all(fused.conv_bn_act_eval_supported(cv, b, a_, cur) for cv, b, a_ in units):
                for j, (cv, b, a_) in enumerate(units):
                    last = j == len(units) - 1
                    cur = fused.conv_bn_act_eval(cv, b, a_, cur, want32=last or not torch.is_autocast_enabled(),
                                                 want16=not last and torch.is_autocast_enabled())
                return cur
        # channels-last convolutions hand their output over as a token-major matrix: BN (+ GELU) on the blocks' own kernels
        if first is not None or (fused.STEM_BN and fused.ENABLED and x.is_cuda and x.dtype == torch.float32
                                 and not torch.is_autocast_enabled()):
            if first is None:
                x, i = x.contiguous(memory_format=torch.channels_last), 0
            else:
                x, i = first, start
            while i < len(mods):
                if isinstance(mods[i], nn.Conv2d):
                    x = fused.conv_before_bn(mods[i], mods[i + 1], x) if i + 1 < len(mods) else mods[i](x)
                    i += 1
                    continue
                bn = mods[i]
                act = mods[i + 1] if i + 1 < len(mods) and not isinstance(mods[i + 1], nn.Conv2d) else None
                own = x.dtype == torch.float32 and not torch.is_autocast_enabled() and fused.STEM_BN
                x = fused.bn_act(x, bn, act) if own and fused.bn_act_supported(bn, x, act) else (bn(x) if act is None else act(bn(x)))
                i += 1 if act is None else 2
            return x
        return self.convs(x)


class Downsample(nn.Module):
    """3x3 stride-2 conv + BN between stages (reference gkgnet.py:103-118)."""

    def __init__(self, in_dim=3, out_dim=768):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(in_dim, out_dim, 3, stride=2, padding=1), build_norm(out_dim))

    def forward(self, x):
        if fused.conv_bn_act_eval_supported(self.conv[0], self.conv[1], None, x):
            # inference: convolution + ONE pass (eval BN + bias) -> fp32 channels-last, its bf16 rounding riding along under autocast
            return fused.conv_bn_act_eval(self.conv[0], self.conv[1], None, x, want32=True, want16=torch.is_autocast_enabled())
        if fused.STEM_BN and fused.ENABLED and x.is_cuda and x.dtype == torch.float32 and not torch.is_autocast_enabled():
            y = fused.conv_before_bn(self.conv[0], self.conv[1], x.contiguous(memory_format=torch.channels_last))
            if fused.bn_act_supported(self.conv[1], y, None):
                return fused.bn_act(y, self.conv[1], None)
            return self.conv[1](y)
        return self.conv(x)


def _arch(blocks, channels):
    return dict(k=9, conv="mr", act="gelu", norm="batch", bias=True, dropout=0.0, use_dilation=True, epsilon=0.2,
                use_stochastic=False, blocks=blocks, channels=channels, emb_dims=1024)


@BACKBONES.register_module()
class GKGNet(nn.Module):
    arch_settings = {
        "t": _arch([2, 2, 6, 2], [48, 96, 240, 384]),
        "s": _arch([2, 2, 6, 2], [80, 160, 400, 640]),
        "m": _arch([2, 2, 16, 2], [96, 192, 384, 768]),       # not in the reference (Pyramid-ViG-M)
    }

    def __init__(self, choice="s", k=9, k_label_gcn=9, use_multi_group=True, backbone_multi_group=True, num_group=2,
                 drop_path=0.0, n_classes=1000, out_indices=(3,), size=576, num_gcn=1, pretrain_path=None,
                 init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg
        opt = self.arch_settings[choice]
        act, norm, bias = opt["act"], opt["norm"], opt["bias"]
        epsilon, stochastic, conv = opt["epsilon"], opt["use_stochastic"], opt["conv"]
        blocks, channels = opt["blocks"], opt["channels"]
        self.n_blocks = sum(blocks)
        reduce_ratios = [4, 2, 1, 1]
        dpr = [v.item() for v in torch.linspace(0, drop_path, self.n_blocks)]     # stochastic depth decay
        max_dilation = 49 // k

        self.register_buffer("label_input", torch.arange(n_classes).view(1, -1), persistent=False)
        self.label_lt = nn.Embedding(n_classes, channels[0], padding_idx=None)
        # index (in self.backbone) of the last block of every stage: label graph convs run there
        self.layer_index = [int(np.sum(blocks[:i + 1]) + i - 1) for i in range(len(blocks))]
        self.out_indices = [int(np.sum(blocks[:i + 1]) + i - 1) for i in out_indices]

        self.stem = Stem(out_dim=channels[0], act=act)
        self.pos_embed = nn.Parameter(torch.zeros(1, channels[0], size // 4, size // 4))
        HW = size // 4 * size // 4

        backbone, gcn_label, ffn_label = [], [], []
        idx = 0
        for i in range(len(blocks)):
            def label_block():
                return GrapherLabel(channels[i], k_label_gcn, 1, "mr", act, norm, bias, stochastic, epsilon,
                                    reduce_ratios[i], n=HW, drop_path=dpr[idx], relative_pos=False,
                                    num_nodes=n_classes, use_multi_group=use_multi_group, num_group=num_group)
            if i < len(blocks) - 1:
                gcn_label.append(nn.Sequential(label_block()))
                ffn_label.append(nn.Sequential(nn.Linear(channels[i], channels[i + 1])))
            else:
                gcn_label.append(nn.ModuleList(label_block() for _ in range(num_gcn)))
            if i > 0:
                backbone.append(Downsample(channels[i - 1], channels[i]))
                HW = HW // 4
            for _ in range(blocks[i]):
                backbone.append(nn.Sequential(
                    Grapher(channels[i], k, min(idx // 4 + 1, max_dilation), conv, act, norm, bias, stochastic,
                            epsilon, reduce_ratios[i], n=HW, drop_path=dpr[idx], relative_pos=True,
                            use_multi_group=backbone_multi_group, num_group=num_group),
                    FFN(channels[i], channels[i] * 4, act=act, drop_path=dpr[idx])))
                idx += 1
        self.backbone = nn.Sequential(*backbone)
        self.gcn_label = nn.Sequential(*gcn_label)
        self.ffn_label = nn.Sequential(*ffn_label)
        self.gap = nn.AdaptiveAvgPool2d((1, 1))
        self.model_init()

    def model_init(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
                m.weight.requires_grad = True
                if m.bias is not None:
                    m.bias.data.zero_()
                    m.bias.requires_grad = True

    def init_weights(self):
        """Checkpoint loading is left to the caller (mmcv's ``init_cfg=dict(type='Pretrained', ...)`` or
        ``load_state_dict``); keys match the reference's."""
        return None

    def forward(self, inputs):
        # (B, n_classes, C1) label queries: label_input is arange(n_classes), so the embedding lookup of the reference
        # (gkgnet.py:218) is the weight itself, broadcast over the batch — its backward is then a batch sum instead of the
        # library's sort-based embedding backward (153 us per step at B = 32)
        if getattr(self, "_lt_identity", None) is None:
            li = self.label_input.reshape(-1)
            self._lt_identity = bool(self.label_lt.max_norm is None and self.label_lt.padding_idx is None
                                     and li.numel() == self.label_lt.num_embeddings
                                     and torch.equal(li.cpu(), torch.arange(li.numel())))
        if self._lt_identity:
            labels = self.label_lt.weight.unsqueeze(0).expand(inputs.size(0), -1, -1)
        else:
            labels = self.label_lt(self.label_input.to(inputs.device).repeat(inputs.size(0), 1))
        x = fused.add_pos_embed(self.stem(inputs), self.pos_embed)
        stage = 0
        edge_index = None
        # the fused blocks pass channels-last activations through without transposing (fused.CHANNELS_LAST): convert once
        # here and after each downsample (a no-op when the convolution already returned channels-last)
        to_cl = fused.CHANNELS_LAST and fused.ENABLED and x.is_cuda
        for i, block in enumerate(self.backbone):
            if to_cl and isinstance(block, (Downsample,)):
                x = block(x)
                x = x.to(dtype=torch.float32, memory_format=torch.channels_last)
                continue
            if to_cl and i == 0:
                x = x.to(dtype=torch.float32, memory_format=torch.channels_last)
            x = block(x)
            if i in self.layer_index:
                for gl in self.gcn_label[stage]:
                    labels, edge_index = gl(labels, x)
                if stage < 3:
                    labels = self.ffn_label[stage](labels)
                stage += 1
        return labels, torch.flatten(self.gap(x), 1), edge_index


register_with_mmcls(GKGNet)


def load_checkpoint(model: nn.Module, path_or_state, prefix: str = "", strict: bool = False):
    """Load a reference checkpoint by key: ``pvig_s_82.1.pth.tar`` (plain ViG backbone weights) or a
    GKGNet-576 ``.pth`` written by mmcv (``{'state_dict': {'backbone.*': ...}}``).  Handles the ``state_dict`` nesting,
    an optional key ``prefix`` (``'backbone.'``; auto-detected when empty) and tensors whose spatial size differs only
    in ``pos_embed`` / ``relative_pos`` (kept from the freshly built model, like mmcv's non-strict load).
    Returns (missing_keys, unexpected_keys, skipped_shape_mismatch)."""
    state = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, str) else path_or_state
    for key in ("state_dict", "model"):
        if isinstance(state, dict) and key in state and isinstance(state[key], dict):
            state = state[key]
    own = model.state_dict()
    if not prefix and not any(k in own for k in state) and any(k.startswith("backbone.") for k in state):
        prefix = "backbone."
    if prefix:
        state = {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}
    skipped = [k for k, v in state.items() if k in own and tuple(own[k].shape) != tuple(v.shape)]
    state = {k: v for k, v in state.items() if k not in skipped}
    result = model.load_state_dict(state, strict=False)
    missing = [k for k in result.missing_keys if k not in skipped]
    if strict and (missing or result.unexpected_keys):
        raise RuntimeError(f"checkpoint mismatch: missing {missing[:5]}, unexpected {result.unexpected_keys[:5]}")
    return missing, list(result.unexpected_keys), skipped
