"""Multi-label mean average precision: the metric behind the reference's only published number
(87.65 mAP, README.md:83).  Same definition as mmcls/core/evaluation/mean_ap.py:6-74: per class, sort by score,
AP = mean over positives of precision at that rank, with label -1 ("difficult") excluded from the ranking; mAP is
the class mean in percent.  Vectorised over classes with numpy."""
from __future__ import annotations

import numpy as np


def average_precision(pred: np.ndarray, target: np.ndarray) -> float:
    eps = np.finfo(np.float32).eps
    order = np.argsort(-pred)
    t = target[order]
    pos = t == 1
    tp = np.cumsum(pos)
    counted = np.cumsum(t != -1)
    precision = np.where(pos, tp / np.maximum(counted, eps), 0.0)
    return float(precision.sum() / np.maximum(tp[-1], eps))


def mAP(pred, target) -> float:
    """pred, target (N, C) arrays or tensors; target in {1, 0, -1}.  Returns mean AP over classes * 100."""
    pred = np.asarray(pred.detach().cpu() if hasattr(pred, "detach") else pred)
    target = np.asarray(target.detach().cpu() if hasattr(target, "detach") else target)
    if pred.shape != target.shape:
        raise ValueError("pred and target should be in the same shape.")
    return float(np.mean([average_precision(pred[:, k], target[:, k]) for k in range(pred.shape[1])]) * 100.0)
