"""COCO annotation reader + evaluation metrics against the reference's own dataset class (fixture F16,
tools/gen_golden.py::coco_case: mmcls/datasets/coco.py run on a synthetic pickled annotation file + random scores)."""
import numpy as np

from util import load_fixture


def test_coco_annotation_reader_and_metrics(tmp_path):
    from gkgnet_amd.coco import CLASSES, coco_metrics, gt_label_matrix, load_coco_annotations
    meta, a = load_fixture("f16_coco")
    ann = tmp_path / "val_test.data"
    ann.write_bytes(a["ann_file_bytes"].tobytes())            # the very file the reference's COCO.load_annotations read
    infos = load_coco_annotations(str(ann), meta["data_prefix"])
    assert len(infos) == meta["N"] and len(CLASSES) == meta["C"] == 80
    assert [i["img_info"]["filename"] for i in infos] == meta["filenames"]
    assert all(i["img_prefix"] == meta["data_prefix"] and i["gt_label"].dtype == np.int8 for i in infos)
    gt = gt_label_matrix(infos)
    assert np.array_equal(gt, a["gt"])
    got = coco_metrics(gt, a["preds"], threshold=0.5)
    assert set(got) == set(meta["metrics"])
    for k, v in meta["metrics"].items():
        assert abs(got[k] - v) <= 1e-9 * max(1.0, abs(v)), (k, got[k], v)
