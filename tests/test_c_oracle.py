"""Pin the C oracle (oracle/gkg_oracle.c — the exact-arithmetic contract the HIP kernels follow)
against the reference's golden vectors.  CPU only."""
import numpy as np
import pytest

from oracle import c_oracle as O
from util import check_indices, load_fixture

OP_CASES = ["op_self_relpos", "op_xy_norelpos", "op_xy_relpos_dil", "op_self_bf16", "op_label_like"]


@pytest.mark.parametrize("name", OP_CASES)
def test_c_oracle_ops_match_reference(name):
    meta, a = load_fixture(name)
    x, y, rp = a["x"], a.get("y"), a.get("relpos")
    idx, center, dist = O.knn(x, y, rp, meta["k"], meta["dilation"], want_dist=True)
    swaps = check_indices(idx, a["edge_index"][0], a["topd"], a["topi"], meta["dilation"])
    assert swaps <= 2
    assert np.array_equal(center, a["edge_index"][1])
    # the contract's distances agree with the reference's matmul-based ones to fp32 rounding
    kd = meta["k"] * meta["dilation"]
    mine = np.sort(dist, axis=-1)[..., :kd]
    assert np.allclose(mine, a["topd"][..., :kd], atol=2e-6)
    ref_idx = a["edge_index"][0].astype(np.int64)
    m, arg = O.mr_fwd(x, y, ref_idx)
    assert np.array_equal(m, a["m"])                       # same fp32 subtract/max -> bit exact
    gx, gsrc = O.mr_bwd(a["gcot"], ref_idx, arg, None if y is None else y.shape[2])
    assert np.allclose(gx, a["dx"], atol=1e-5)
    if y is not None:
        assert np.allclose(gsrc, a["dy"], atol=1e-5)


@pytest.mark.parametrize("name", ["f1_grapher_cfg1", "f2_grapher_g4", "f3_grapher_dil3", "f4a_grapher_r2",
                                  "f4b_grapher_r4", "f7_grapher_bf16in"])
def test_c_oracle_on_grapher_knn_inputs(name):
    """k-NN + max-relative on the exact tensor the reference's graph_conv saw (fc1 output)."""
    import torch
    import torch.nn.functional as F
    meta, a = load_fixture(name)
    B, C, n, r = meta["B"], meta["C"], meta["n"], meta["r"]
    G = meta["G"] if meta["use_multi_group"] else 1
    h1 = torch.from_numpy(a["knn_in"])
    x = h1.reshape(B * G, C // G, n).numpy()
    y = None
    if r > 1:
        y = F.avg_pool2d(h1, r, r).reshape(B * G, C // G, -1).numpy()
    idx, center = O.knn(x, y, a["sd/relative_pos"], meta["k"], meta["dilation"])
    check_indices(idx, a["edge_index"][0], a["topd"], a["topi"], meta["dilation"])
    m, _ = O.mr_fwd(x, y, a["edge_index"][0].astype(np.int64))
    assert np.array_equal(m.reshape(B, C, n), a["m"])


def test_c_oracle_integer_known_answer():
    """F8: exact integer arithmetic (no normalisation): zero tolerance, incl. the distance values."""
    meta, a = load_fixture("f8_integer_kat")
    idx, center, dist = O.knn(a["x"], a["y"], None, meta["k"], 1, normalize=False, want_dist=True)
    assert np.array_equal(idx, a["edge_xy"][0]) and np.array_equal(center, a["edge_xy"][1])
    assert np.array_equal(dist, a["dist_xy"])
    idx, center = O.knn(a["y"], None, None, meta["k"], 1, normalize=False)
    assert np.array_equal(idx, a["edge_self"][0])


def test_c_oracle_tie_rule_hand_derived():
    """Tied instance whose expectation is hand-derived from the documented rule (smaller index first).
    4 keys at the same point, 2 further away: query at origin, k=3, d=1 -> [0,1,2]; d=2 -> [0,2,4]."""
    y = np.zeros((1, 2, 6), np.float32)
    y[0, 0] = [1, 1, 1, 1, 5, 7]
    x = np.zeros((1, 2, 1), np.float32)
    idx, _ = O.knn(x, y, None, 3, 1, normalize=False)
    assert idx.tolist() == [[[0, 1, 2]]]
    idx, _ = O.knn(x, y, None, 3, 2, normalize=False)
    assert idx.tolist() == [[[0, 2, 4]]]
    # argmax tie rule of the aggregation: first neighbour attaining the max
    src = np.array([[[2.0, 2.0, 1.0]]], np.float32)
    m, arg = O.mr_fwd(np.zeros((1, 1, 1), np.float32), src, np.array([[[2, 1, 0]]], np.int64))
    assert m.item() == 2.0 and arg.item() == 1
