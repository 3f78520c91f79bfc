"""Tiny stand-in for the mmcv registry used by the reference (mmcls/models/builder.py:6-19):
``BACKBONES.register_module()`` + ``build_backbone(dict(type='GKGNet', ...))``.  When mmcls itself is
importable the class is ALSO registered there, so ``configs/gkgnet/gkgnet_coco_576.py`` builds this
implementation unchanged."""
from __future__ import annotations


class Registry:
    def __init__(self, name: str):
        self.name = name
        self._modules = {}

    def register_module(self, name=None, force=False, module=None):
        def _register(cls):
            key = name or cls.__name__
            if key in self._modules and not force:
                raise KeyError(f"{key} is already registered in {self.name}")
            self._modules[key] = cls
            return cls
        if module is not None:
            return _register(module)
        return _register

    def get(self, key):
        return self._modules.get(key)

    def build(self, cfg: dict):
        cfg = dict(cfg)
        typ = cfg.pop("type")
        cls = self.get(typ) if isinstance(typ, str) else typ
        if cls is None:
            raise KeyError(f"{typ} is not in the {self.name} registry")
        return cls(**cfg)


MODELS = Registry("models")
BACKBONES = MODELS


def build_backbone(cfg: dict):
    return BACKBONES.build(cfg)


def register_with_mmcls(cls):
    """Best effort: expose ``cls`` to a real mmcls installation (force=True replaces the reference class)."""
    try:
        from mmcls.models.builder import BACKBONES as MM_BACKBONES  # type: ignore
        MM_BACKBONES.register_module(name=cls.__name__, force=True, module=cls)
    except Exception:
        pass
    return cls
