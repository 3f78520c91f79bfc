"""Pin the CPU oracle (oracle/torch_ref.py) against golden vectors produced by the reference itself
(tools/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from util import check_indices, grads_from, load_fixture, state_from

GRAPHER_CASES = ["f1_grapher_cfg1", "f2_grapher_g4", "f3_grapher_dil3", "f4a_grapher_r2", "f4b_grapher_r4",
                 "f7_grapher_bf16in", "f11_grapher_edgeconv"]
LABEL_CASES = ["f5_label_g2", "f5b_label_g1"]
OP_CASES = ["op_self_relpos", "op_xy_norelpos", "op_xy_relpos_dil", "op_self_bf16", "op_label_like"]


def _t(a):
    return torch.from_numpy(np.array(a))


@pytest.mark.parametrize("name", GRAPHER_CASES)
def test_grapher_matches_reference(name):
    meta, a = load_fixture(name)
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k and k != "relative_pos")
         for k, v in state_from(a).items()}
    groups = meta["G"] if meta["use_multi_group"] else 1
    kw = dict(k=meta["k"], dilation=meta["dilation"], r=meta["r"], groups=groups, conv=meta["conv"])
    x = _t(a["x"])
    with torch.no_grad():
        out_eval, aux_e = R.grapher_forward(x, p, training=False, return_aux=True, **kw)
    assert torch.allclose(out_eval, _t(a["out_eval"]), atol=2e-5, rtol=1e-5)
    check_indices(aux_e["edge_index"][0].numpy(), a["edge_index_eval"][0], *_topd_eval(R, aux_e, p, meta), meta["dilation"])

    xg = x.clone().requires_grad_(True)
    out, aux = R.grapher_forward(xg, p, training=True, return_aux=True, **kw)
    swaps = check_indices(aux["edge_index"][0].numpy(), a["edge_index"][0], a["topd"], a["topi"], meta["dilation"])
    assert np.array_equal(aux["edge_index"][1].numpy(), a["edge_index"][1])
    assert torch.allclose(aux["fc1"], _t(a["knn_in"]), atol=1e-5, rtol=1e-5)
    if swaps == 0:
        if meta["conv"] == "mr":
            m = aux["m"].reshape(meta["B"], meta["C"], -1)
            assert torch.allclose(m, _t(a["m"]), atol=1e-5, rtol=1e-5)
        assert torch.allclose(out, _t(a["out"]), atol=5e-5, rtol=1e-4)
        (out * _t(a["cot"])).sum().backward()
        assert torch.allclose(xg.grad, _t(a["dx"]), atol=2e-4, rtol=1e-3)
        for k, g in grads_from(a).items():
            assert torch.allclose(p[k].grad, g, atol=3e-4, rtol=2e-3), k


def _topd_eval(R, aux, p, meta):
    """Eval-mode run has no stored topd: rebuild it from the oracle's own distances (exactness of the
    eval indices is then checked against the reference's indices with the same near-tie rule)."""
    h1 = aux["fc1"]
    b, c = h1.shape[:2]
    groups = meta["G"] if meta["use_multi_group"] else 1
    xq = h1.reshape(b * groups, c // groups, -1, 1)
    yk = None
    if meta["r"] > 1:
        yk = torch.nn.functional.avg_pool2d(h1, meta["r"], meta["r"]).reshape(b * groups, c // groups, -1, 1)
    dist = R.knn_distances(xq, yk, p.get("relative_pos"))
    kd = min(meta["k"] * meta["dilation"] + 1, dist.shape[-1])
    v, i = torch.topk(-dist, kd)
    return (-v).numpy(), i.numpy()


@pytest.mark.parametrize("name", LABEL_CASES)
def test_grapher_label_matches_reference(name):
    meta, a = load_fixture(name)
    p = {k: v.clone().requires_grad_(v.dtype.is_floating_point and "running" not in k)
         for k, v in state_from(a).items()}
    groups = meta["G"] if meta["use_multi_group"] else 1
    e, feat = _t(a["e"]), _t(a["feat"])
    with torch.no_grad():
        out_eval, idx_eval = R.grapher_label_forward(e, feat, p, k=meta["k"], groups=groups,
                                                     use_multi_group=meta["use_multi_group"], training=False)
    assert idx_eval.shape == a["nn_idx_eval"].shape
    assert torch.allclose(out_eval, _t(a["out_eval"]), atol=2e-5, rtol=1e-5)
    eg, fg = e.clone().requires_grad_(True), feat.clone().requires_grad_(True)
    out, idx, aux = R.grapher_label_forward(eg, fg, p, k=meta["k"], groups=groups,
                                            use_multi_group=meta["use_multi_group"], training=True, return_aux=True)
    assert idx.shape == a["nn_idx"].shape          # (BG,L,k) multi-group, (2,B,L,k) single-group quirk
    nn_mine = idx.numpy() if meta["use_multi_group"] else idx[0].numpy()
    nn_ref = a["nn_idx"] if meta["use_multi_group"] else a["nn_idx"][0]
    swaps = check_indices(nn_mine, nn_ref, a["topd"], a["topi"])
    assert torch.allclose(aux["fc1"], _t(a["knn_in"]), atol=1e-5, rtol=1e-5)
    if swaps == 0:
        assert torch.allclose(aux["m"].reshape(meta["B"], meta["C"], -1), _t(a["m"]), atol=1e-5, rtol=1e-5)
        assert torch.allclose(out, _t(a["out"]), atol=5e-5, rtol=1e-4)
        (out * _t(a["cot"])).sum().backward()
        assert torch.allclose(eg.grad, _t(a["de"]), atol=2e-4, rtol=1e-3)
        assert torch.allclose(fg.grad, _t(a["dfeat"]), atol=2e-4, rtol=1e-3)
        for k, g in grads_from(a).items():
            assert torch.allclose(p[k].grad, g, atol=3e-4, rtol=2e-3), k


@pytest.mark.parametrize("name", OP_CASES)
def test_knn_and_max_relative_ops(name):
    meta, a = load_fixture(name)
    x = _t(a["x"]).unsqueeze(-1)
    y = _t(a["y"]).unsqueeze(-1) if "y" in a else None
    rp = _t(a["relpos"]) if "relpos" in a else None
    edge = R.knn_graph(x, y, rp, meta["k"], meta["dilation"])
    swaps = check_indices(edge[0].numpy(), a["edge_index"][0], a["topd"], a["topi"], meta["dilation"])
    assert np.array_equal(edge[1].numpy(), a["edge_index"][1])
    ref_edge = _t(a["edge_index"]).long()
    xg = x.clone().requires_grad_(True)
    yg = None if y is None else y.clone().requires_grad_(True)
    m = R.max_relative(xg, ref_edge[0], yg)
    assert torch.equal(m, _t(a["m"]))                      # same fp32 ops -> bit exact
    (m * _t(a["gcot"])).sum().backward()
    assert torch.allclose(xg.grad.squeeze(-1), _t(a["dx"]), atol=1e-6)
    if y is not None:
        assert torch.allclose(yg.grad.squeeze(-1), _t(a["dy"]), atol=1e-6)
    assert swaps <= 2


def test_integer_known_answer():
    """F8: exact integer arithmetic, no normalisation — indices must match with no tolerance."""
    meta, a = load_fixture("f8_integer_kat")
    x, y = _t(a["x"]).unsqueeze(-1), _t(a["y"]).unsqueeze(-1)
    e_xy = R.knn_graph(x, y, None, meta["k"], 1, normalize=False)
    assert np.array_equal(e_xy.numpy(), a["edge_xy"])
    e_self = R.knn_graph(y, None, None, meta["k"], 1, normalize=False)
    assert np.array_equal(e_self.numpy(), a["edge_self"])
    assert np.array_equal(R.knn_distances(x, y, None, normalize=False).numpy(), a["dist_xy"])


def test_relative_pos_constants():
    """F9: the frozen positional bias, incl. the flattened-axis bicubic quirk for r>1."""
    meta, a = load_fixture("f9_relpos")
    for C, n, r in meta["combos"]:
        got = R.grapher_relative_pos(C, n, r)
        assert torch.equal(got, _t(a[f"rp_{C}_{n}_{r}"])), (C, n, r)
    base = R.grapher_relative_pos(32, 64, 2)
    got = R.runtime_relative_pos(base, 64, 2, 10, 10)
    assert torch.allclose(got, _t(a["rp_runtime_32_64_2_to_10x10"]), atol=1e-6)


def test_amp_fixture_fp32_leg_matches_the_oracle():
    """F17's fp32 leg (Grapher -> GrapherLabel, train mode, forward + backward) through the oracle: pins the fixture's wiring
    (weights, cotangents, the label chain) on the CPU side; the fp16-autocast leg is the GPU test's bar
    (tests/test_hip_autocast_trainstep.py)."""
    meta, a = load_fixture("f17_amp_fp16")
    pg = {k_[len("g/sd/"):]: _t(v) for k_, v in a.items() if k_.startswith("g/sd/")}
    pl = {k_[len("gl/sd/"):]: _t(v) for k_, v in a.items() if k_.startswith("gl/sd/")}
    x = _t(a["x"]).requires_grad_(True)
    e = _t(a["e"]).requires_grad_(True)
    out = R.grapher_forward(x, pg, k=meta["k"], dilation=meta["dilation"], r=1, groups=meta["G"], training=True)
    e2, idx = R.grapher_label_forward(e, out, pl, k=meta["k"], groups=meta["G"], training=True)
    ((out * _t(a["cot_x"])).sum() + (e2 * _t(a["cot_e"])).sum()).backward()
    assert torch.allclose(out, _t(a["fp32/out"]), atol=1e-4, rtol=1e-4)
    assert torch.allclose(e2, _t(a["fp32/labels"]), atol=1e-4, rtol=1e-4)
    assert np.array_equal(idx.numpy(), a["fp32/idx"])
    assert torch.allclose(x.grad, _t(a["fp32/dx"]), atol=1e-5, rtol=1e-3)
    assert torch.allclose(e.grad, _t(a["fp32/de"]), atol=1e-5, rtol=1e-3)
