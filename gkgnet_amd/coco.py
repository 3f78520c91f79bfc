"""COCO multi-label annotations and the evaluation the reference runs on them.

    load_coco_annotations   reference mmcls/datasets/coco.py:261-285  (COCO.load_annotations: a pickled list of
                            {'file_name', 'objects' (80-vector)} records -> per-image info dicts)
    coco_metrics            reference mmcls/datasets/coco.py:65-176   (average_precision / mAP / get_coco_metrics: mAP with
                            the dataset's own AP definition, CP / CR / CF1 / OP / OR / OF1 at threshold 0.5 and their
                            top-3 variants; scikit-learn's precision / recall restated with numpy, zero_division = 0)
Plain numpy: evaluation is host work on (N, 80) score matrices."""
from __future__ import annotations

import pickle

import numpy as np

CLASSES = (
    "person", "bicycle", "car", "motorcycle", "airplane", "bus", "train", "truck", "boat", "traffic light", "fire hydrant",
    "stop sign", "parking meter", "bench", "bird", "cat", "dog", "horse", "sheep", "cow", "elephant", "bear", "zebra",
    "giraffe", "backpack", "umbrella", "handbag", "tie", "suitcase", "frisbee", "skis", "snowboard", "sports ball", "kite",
    "baseball bat", "baseball glove", "skateboard", "surfboard", "tennis racket", "bottle", "wine glass", "cup", "fork",
    "knife", "spoon", "bowl", "banana", "apple", "sandwich", "orange", "broccoli", "carrot", "hot dog", "pizza", "donut",
    "cake", "chair", "couch", "potted plant", "bed", "dining table", "toilet", "tv", "laptop", "mouse", "remote",
    "keyboard", "cell phone", "microwave", "oven", "toaster", "sink", "refrigerator", "book", "clock", "vase", "scissors",
    "teddy bear", "hair drier", "toothbrush")


def load_coco_annotations(ann_file: str, data_prefix: str = ""):
    """The reference's ``.data`` annotation file: a pickled sequence of records with 'file_name' and 'objects' (multi-hot
    class vector).  Returns the list of info dicts the reference's dataset keeps (img_prefix, img_info.filename,
    gt_label int8)."""
    with open(ann_file, "rb") as fh:
        records = pickle.load(fh)
    infos = []
    for rec in records:
        infos.append(dict(img_prefix=data_prefix, img_info=dict(filename=rec["file_name"]),
                          gt_label=np.asarray(rec["objects"]).astype(np.int8)))
    return infos


def gt_label_matrix(infos) -> np.ndarray:
    return np.stack([i["gt_label"] for i in infos])


def average_precision(output: np.ndarray, target: np.ndarray) -> float:
    """AP of one class as the COCO dataset class computes it: descending scores, mean of precision@i over the positives
    (eps 1e-8 in the denominator; unlike core/evaluation/mean_ap.py there is no 'difficult' label here)."""
    order = output.argsort()[::-1]
    t = target[order] == 1
    pos = np.cumsum(t).astype(np.float64)
    total = pos[-1] if len(pos) else 0.0
    rank = np.arange(1, len(output) + 1, dtype=np.float64)
    return float(np.sum(np.where(t, pos / rank, 0.0)) / (total + 1e-8))


def _prf(targets: np.ndarray, pred01: np.ndarray):
    """macro / micro precision and recall of binary indicator matrices (zero_division -> 0, like scikit-learn)."""
    t = targets == 1
    p = pred01 == 1
    tp = (t & p).sum(0).astype(np.float64)
    pp = p.sum(0).astype(np.float64)
    tt = t.sum(0).astype(np.float64)
    cp = float(np.mean(np.divide(tp, pp, out=np.zeros_like(tp), where=pp > 0)))
    cr = float(np.mean(np.divide(tp, tt, out=np.zeros_like(tp), where=tt > 0)))
    op = float(tp.sum() / pp.sum()) if pp.sum() > 0 else 0.0
    orr = float(tp.sum() / tt.sum()) if tt.sum() > 0 else 0.0
    return cp, cr, op, orr


def _f1(a: float, b: float) -> float:
    return (2 * a * b) / (a + b) if (a + b) > 0 else float("nan")


def coco_metrics(targets, preds, threshold: float = 0.5) -> dict:
    """targets (N, C) in {0, 1}, preds (N, C) scores in [0, 1] -> the reference's metric dictionary (fractions, not %)."""
    targets = np.asarray(targets)
    preds = np.asarray(preds, dtype=np.float64) if np.asarray(preds).dtype == np.float64 else np.asarray(preds)
    ap = np.array([average_precision(preds[:, k], targets[:, k]) for k in range(preds.shape[1])])
    out = {"mAP": float(ap.mean())}
    top3 = np.sort(preds)[:, -3].reshape(-1, 1)
    p3 = preds.copy()
    p3[p3 < top3] = 0
    p3 = (p3 >= threshold).astype(np.int8)
    pt = (preds >= threshold).astype(np.int8)
    cp, cr, op, orr = _prf(targets, pt)
    out.update(CP=cp, CR=cr, CF1=_f1(cp, cr), OP=op, OR=orr, OF1=_f1(op, orr))
    cp, cr, op, orr = _prf(targets, p3)
    out.update(CP_top3=cp, CR_top3=cr, OP_top3=op, CF1_top3=_f1(cp, cr), OR_top3=orr, OF1_top3=_f1(op, orr))
    return out
