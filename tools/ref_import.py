"""Load the reference's Group-KNN hot-path files in THIS container (build-time only).

The upstream tree under /root/reference is read-only and cannot be imported as
``mmcls`` (it asserts an mmcv version, and mmcv / timm / easydict are absent).
This helper injects minimal stand-ins for those third-party names into
``sys.modules`` and executes the four ``vig_model`` files (+ ``gkgnet.py``) by
path, so the *reference's own code* produces the golden vectors kept under
``tests/golden/``.  Nothing here travels to the GPU box: only the generated
fixtures do.  It contains no reference source.

Recipe: SURVEY.md Appendix A.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF_ROOT = os.environ.get("GKG_REFERENCE_ROOT", "/root/reference")


class _DropPath(nn.Module):
    """Standard stochastic depth (identity when p == 0 or in eval mode)."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep


class _EasyDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = _EasyDict(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:  # abc machinery probes attributes
            raise AttributeError(k) from e

    __setattr__ = dict.__setitem__


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


def _build_norm_layer(cfg, num_features, postfix=""):
    kinds = {"BN": nn.BatchNorm2d, "SyncBN": nn.SyncBatchNorm}
    return "bn" + str(postfix), kinds[cfg["type"]](num_features)


class _Registry:
    """Records registered classes by name so that build_loss(dict(type=...)) works."""

    def __init__(self):
        self.classes = {}

    def register_module(self, *a, **k):
        def deco(cls):
            self.classes[cls.__name__] = cls
            return cls
        return deco

    def build(self, cfg):
        cfg = dict(cfg)
        return self.classes[cfg.pop("type")](**cfg)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _exec(name, relpath, search=None):
    path = os.path.join(REF_ROOT, relpath)
    spec = importlib.util.spec_from_file_location(
        name, path, submodule_search_locations=search)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


_LOADED = None


def load_reference(with_backbone: bool = True, with_head: bool = False):
    """Returns a namespace with the reference's vig_model (and gkgnet) modules."""
    global _LOADED
    if _LOADED is not None and (not with_head or hasattr(_LOADED, "head")):
        return _LOADED
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    if not hasattr(np, "float"):
        np.float = float  # pos_embed.py:74 uses the removed alias

    _mod("mmcv")
    _mod("mmcv.cnn", build_norm_layer=_build_norm_layer, ConvModule=None,
         build_conv_layer=None, constant_init=None)
    _mod("mmcv.cnn.bricks", DropPath=_DropPath)
    _mod("mmcv.runner", BaseModule=_BaseModule)
    _mod("easydict", EasyDict=_EasyDict)
    _mod("timm")
    _mod("timm.data", IMAGENET_DEFAULT_MEAN=(0.485, 0.456, 0.406),
         IMAGENET_DEFAULT_STD=(0.229, 0.224, 0.225))
    _mod("timm.models")
    _mod("timm.models.layers", DropPath=_DropPath)

    _pkg("mmcls")
    _pkg("mmcls.models")
    _pkg("mmcls.models.utils")
    _pkg("mmcls.models.backbones")
    _reg = _Registry()
    _mod("mmcls.models.builder", BACKBONES=_reg, HEADS=_reg, LOSSES=_reg, build_loss=_reg.build)

    base = "mmcls/models"
    _exec("mmcls.models.utils.differentiable_topk", f"{base}/utils/differentiable_topk.py")
    vig_dir = os.path.join(REF_ROOT, base, "backbones/vig_model")
    vig = _exec("mmcls.models.backbones.vig_model",
                f"{base}/backbones/vig_model/__init__.py", search=[vig_dir])
    ns = types.SimpleNamespace(vig=vig,
                               torch_edge=sys.modules["mmcls.models.backbones.vig_model.torch_edge"],
                               torch_nn=sys.modules["mmcls.models.backbones.vig_model.torch_nn"],
                               torch_vertex=sys.modules["mmcls.models.backbones.vig_model.torch_vertex"],
                               pos_embed=sys.modules["mmcls.models.backbones.vig_model.pos_embed"])
    if with_backbone:
        _exec("mmcls.models.backbones.base_backbone", f"{base}/backbones/base_backbone.py")
        ns.gkgnet = _exec("mmcls.models.backbones.gkgnet", f"{base}/backbones/gkgnet.py")
        # gkgnet.py:264 hard-codes .cuda(); neutralise on a CPU-only host.
        if not torch.cuda.is_available():
            torch.Tensor.cuda = lambda self, *a, **k: self
    if with_head:
        # head + losses: heads/label_query_head.py, heads/cls_head.py, losses/{utils,accuracy,
        # cross_entropy_loss,label_smooth_loss,asymmetric_loss}.py
        sys.modules["mmcls.models.utils"].is_tracing = lambda: False
        losses_dir = os.path.join(REF_ROOT, base, "losses")
        lp = _pkg("mmcls.models.losses")
        lp.__path__ = [losses_dir]
        for name in ("utils", "accuracy", "cross_entropy_loss", "label_smooth_loss", "asymmetric_loss"):
            m = _exec(f"mmcls.models.losses.{name}", f"{base}/losses/{name}.py")
            setattr(lp, name, m)
        lp.Accuracy = lp.accuracy.Accuracy
        lp.LabelSmoothLoss = lp.label_smooth_loss.LabelSmoothLoss
        lp.AsymmetricLoss = lp.asymmetric_loss.AsymmetricLoss
        hp = _pkg("mmcls.models.heads")
        hp.__path__ = [os.path.join(REF_ROOT, base, "heads")]
        _exec("mmcls.models.heads.base_head", f"{base}/heads/base_head.py")
        _exec("mmcls.models.heads.cls_head", f"{base}/heads/cls_head.py")
        ns.head = _exec("mmcls.models.heads.label_query_head", f"{base}/heads/label_query_head.py")
        ns.losses = lp
        if not torch.cuda.is_available():
            torch.Tensor.cuda = lambda self, *a, **k: self
    _LOADED = ns
    return ns


if __name__ == "__main__":
    ref = load_reference()
    torch.manual_seed(0)
    g = ref.vig.Grapher(64, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=196,
                        relative_pos=True, use_multi_group=False)
    print(g(torch.randn(2, 64, 14, 14)).shape)
