"""GPU parity tests of the HIP operators (through the C ABI) against the C oracle (bit-exact) and the
reference's golden vectors (near-tie protocol).  Run on the MI355X box: pytest -m gpu."""
import numpy as np
import pytest
import torch

from util import check_indices, load_fixture

pytestmark = pytest.mark.gpu

OP_CASES = ["op_self_relpos", "op_xy_norelpos", "op_xy_relpos_dil", "op_self_bf16", "op_label_like"]


def _dev(a, dtype=torch.float32):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


def _shape_seed(shape, salt=0):
    """Process-independent seed (hash() of a tuple holding None is id-based on CPython 3.10)."""
    import zlib
    return (zlib.crc32(repr(shape).encode()) + salt) & 0xFFFF


def _rand_case(seed, BG, c, N, M, relpos, scale=1.0):
    rng = np.random.RandomState(seed)
    x = (rng.standard_normal((BG, c, N)) * scale).astype(np.float32)
    y = None if M is None else (rng.standard_normal((BG, c, M)) * scale).astype(np.float32)
    rp = None
    if relpos:
        rp = -rng.random_sample((N, N if M is None else M)).astype(np.float32)
    return x, y, rp


# (BG, c, N, M(None=self), k, d, relpos)
KNN_SHAPES = [
    (2, 64, 196, None, 9, 1, True),      # cfg1
    (3, 20, 70, None, 5, 2, True),
    (4, 12, 33, 150, 9, 1, False),
    (2, 48, 100, 25, 4, 3, True),        # M % 4 != 0 with relpos, kd == 12
    (2, 7, 65, 130, 3, 1, True),         # odd c (zero-padded k-pair), M % 4 != 0
    (1, 80, 324, None, 9, 1, True),      # cfg2-literal group problem
    (1, 320, 324, None, 9, 3, True),     # cfg2-ref: c=320, top-27
    (2, 40, 80, 1000, 9, 1, False),      # label-like: few queries, many keys -> split-key path
    (1, 16, 200, 81, 18, 2, True),       # kd = 36
    (1, 8, 50, 70, 32, 2, False),        # kd = 64 (largest list)
    (5, 200, 130, None, 9, 2, True),     # stage-3-like c=200, kd=18
    (1, 4, 40, 30000, 9, 1, False),      # keys beyond the LDS-resident row budget: gather / atomic fallbacks
    (16, 12, 8200, 1100, 18, 1, True),   # pvig_m stage-1-like narrow groups, 2 064 query tiles: single-wave buffered form
    (32, 8, 4100, 700, 9, 1, False),     # the same form with the 9-entry list, ragged last query / key tiles
]


@pytest.mark.parametrize("shape", KNN_SHAPES)
def test_knn_bit_exact_vs_c_oracle(shape):
    from gkgnet_amd import ops
    from oracle import c_oracle as O
    BG, c, N, M, k, d, relpos = shape
    x, y, rp = _rand_case(_shape_seed(shape), BG, c, N, M, relpos)
    want_idx, want_center = O.knn(x, y, rp, k, d)
    edge = ops.knn_graph(_dev(x), _dev(y), None if rp is None else _dev(rp).unsqueeze(0), k, d)
    torch.cuda.synchronize()
    got = edge.cpu().numpy()
    assert got.shape == (2, BG, N, k) and got.dtype == np.int64
    assert np.array_equal(got[0], want_idx)
    assert np.array_equal(got[1], want_center)


@pytest.mark.parametrize("shape", KNN_SHAPES)
def test_max_relative_bit_exact_and_backward(shape):
    from gkgnet_amd import ops
    from oracle import c_oracle as O
    BG, c, N, M, k, d, relpos = shape
    x, y, rp = _rand_case(_shape_seed(shape, 7), BG, c, N, M, False)
    idx, _ = O.knn(x, y, None, k, d)
    want_m, want_arg = O.mr_fwd(x, y, idx)
    xd = _dev(x).requires_grad_(True)
    yd = None if y is None else _dev(y).requires_grad_(True)
    m = ops.max_relative(xd, _dev(idx, torch.int64), yd)
    assert np.array_equal(m.detach().cpu().numpy(), want_m)
    g = np.random.RandomState(5).standard_normal(want_m.shape).astype(np.float32)
    m.backward(_dev(g))
    want_gx, want_gsrc = O.mr_bwd(g, idx, want_arg, None if y is None else y.shape[2])
    assert np.allclose(xd.grad.cpu().numpy(), want_gx, atol=1e-5, rtol=1e-5)
    if y is not None:
        assert np.allclose(yd.grad.cpu().numpy(), want_gsrc, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("name", OP_CASES)
def test_ops_against_reference_golden(name):
    from gkgnet_amd import ops
    meta, a = load_fixture(name)
    x, y, rp = a["x"], a.get("y"), a.get("relpos")
    edge = ops.knn_graph(_dev(x), _dev(y), _dev(rp), meta["k"], meta["dilation"]).cpu().numpy()
    swaps = check_indices(edge[0], a["edge_index"][0], a["topd"], a["topi"], meta["dilation"])
    assert swaps <= 2
    assert np.array_equal(edge[1], a["edge_index"][1])
    xd = _dev(x).requires_grad_(True)
    yd = None if y is None else _dev(y).requires_grad_(True)
    m = ops.max_relative(xd, _dev(a["edge_index"][0], torch.int64), yd)
    assert np.array_equal(m.detach().cpu().numpy(), a["m"])
    m.backward(_dev(a["gcot"]))
    assert np.allclose(xd.grad.cpu().numpy(), a["dx"], atol=1e-5)
    if y is not None:
        assert np.allclose(yd.grad.cpu().numpy(), a["dy"], atol=1e-5)


def test_integer_known_answer_and_tie_rule():
    from gkgnet_amd import ops
    meta, a = load_fixture("f8_integer_kat")
    e = ops.knn_graph(_dev(a["x"]), _dev(a["y"]), None, meta["k"], 1, normalize=False).cpu().numpy()
    assert np.array_equal(e, a["edge_xy"])
    e = ops.knn_graph(_dev(a["y"]), None, None, meta["k"], 1, normalize=False).cpu().numpy()
    assert np.array_equal(e, a["edge_self"])
    # documented tie rule: smaller key index first (hand-derived expectation)
    y = np.zeros((1, 2, 6), np.float32); y[0, 0] = [1, 1, 1, 1, 5, 7]
    x = np.zeros((1, 2, 1), np.float32)
    assert ops.knn_graph(_dev(x), _dev(y), None, 3, 1, normalize=False)[0].cpu().tolist() == [[[0, 1, 2]]]
    assert ops.knn_graph(_dev(x), _dev(y), None, 3, 2, normalize=False)[0].cpu().tolist() == [[[0, 2, 4]]]
    # ties across key tiles / waves / splits: 200 identical keys
    y = np.ones((1, 4, 200), np.float32); x = np.zeros((1, 4, 3), np.float32)
    got = ops.knn_graph(_dev(x), _dev(y), None, 9, 2, normalize=False)[0].cpu().numpy()
    assert np.array_equal(got, np.broadcast_to(np.arange(0, 18, 2), (1, 3, 9)))
    # aggregation argmax tie rule: first neighbour attaining the max gets the gradient
    src = _dev(np.array([[[2.0, 2.0, 1.0]]], np.float32)).requires_grad_(True)
    xz = _dev(np.zeros((1, 1, 1), np.float32)).requires_grad_(True)
    m = ops.max_relative(xz, _dev(np.array([[[2, 1, 0]]]), torch.int64), src)
    m.backward(torch.ones_like(m))
    assert m.item() == 2.0 and src.grad.cpu().tolist() == [[[0.0, 1.0, 0.0]]] and xz.grad.item() == -1.0


def test_bf16_inputs_match_fp32_math_on_rounded_values():
    """bf16 I/O: distances are still accumulated in fp32 -> same graph as fp32 math on the rounded values
    (fixture F7 semantics)."""
    from gkgnet_amd import ops
    from oracle import c_oracle as O
    x, y, rp = _rand_case(77, 2, 40, 96, 150, False)
    xb = torch.from_numpy(x).to(torch.bfloat16)
    yb = torch.from_numpy(y).to(torch.bfloat16)
    want_idx, _ = O.knn(xb.float().numpy(), yb.float().numpy(), None, 9, 1)
    got = ops.knn_graph(xb.cuda(), yb.cuda(), None, 9, 1)[0].cpu().numpy()
    assert np.array_equal(got, want_idx)
    want_m, _ = O.mr_fwd(xb.float().numpy(), yb.float().numpy(), want_idx)
    m = ops.max_relative(xb.cuda(), torch.from_numpy(want_idx).cuda(), yb.cuda())
    assert m.dtype == torch.bfloat16
    assert np.array_equal(m.float().cpu().numpy(), torch.from_numpy(want_m).to(torch.bfloat16).float().numpy())


def test_errors_are_loud():
    from gkgnet_amd import _lib, ops
    with pytest.raises(_lib.GkgError):
        ops.knn_graph(torch.zeros(1, 4, 8), None, None, 3, 1)               # CPU tensor: no fallback
    with pytest.raises(_lib.GkgError):
        ops.knn_graph(torch.zeros(1, 4, 8, device="cuda"), None, None, 9, 1)  # k > M
    with pytest.raises(_lib.GkgError):
        ops.knn_graph(torch.zeros(1, 4, 8, device="cuda", dtype=torch.float64), None, None, 3, 1)


def test_fp16_inputs_match_fp32_math_on_rounded_values():
    """fp16 I/O (the reference trains under fp16 AMP, configs/gkgnet/gkgnet_coco_576.py:146): features are widened
    exactly and distances accumulated in fp32 -> the same graph / aggregation as fp32 math on the fp16-rounded values,
    bit for bit against the C oracle; the backward within fp16 rounding."""
    from gkgnet_amd import ops
    from oracle import c_oracle as O
    for seed, (BG, c, N, M, k, d) in enumerate([(2, 40, 96, 150, 9, 1), (3, 24, 130, None, 6, 2), (1, 16, 80, 3000, 9, 1)]):
        x, y, _ = _rand_case(90 + seed, BG, c, N, M, False)
        xh = torch.from_numpy(x).to(torch.float16)
        yh = None if y is None else torch.from_numpy(y).to(torch.float16)
        xo = xh.float().numpy()
        yo = None if yh is None else yh.float().numpy()
        want_idx, want_center = O.knn(xo, yo, None, k, d)
        edge = ops.knn_graph(xh.cuda(), None if yh is None else yh.cuda(), None, k, d).cpu().numpy()
        assert np.array_equal(edge[0], want_idx) and np.array_equal(edge[1], want_center)
        want_m, want_arg = O.mr_fwd(xo, yo, want_idx)
        xg = xh.cuda().requires_grad_(True)
        yg = None if yh is None else yh.cuda().requires_grad_(True)
        m = ops.max_relative(xg, torch.from_numpy(want_idx).cuda(), yg)
        assert m.dtype == torch.float16
        assert np.array_equal(m.detach().float().cpu().numpy(), torch.from_numpy(want_m).to(torch.float16).float().numpy())
        if M is None or M < 2000:               # (the > 24k-key fallback scatters with fp32 atomics: fp32 tensors only)
            g = torch.from_numpy(np.random.RandomState(7).standard_normal(want_m.shape).astype(np.float32)).to(torch.float16)
            m.backward(g.cuda())
            gx, gs = O.mr_bwd(g.float().numpy(), want_idx, want_arg, M)
            assert np.allclose(xg.grad.float().cpu().numpy(), gx, atol=2e-2, rtol=2e-3)
            if M is not None:
                assert np.allclose(yg.grad.float().cpu().numpy(), gs, atol=2e-2, rtol=2e-3)


def _set_mode(monkeypatch, select):
    """direct / buffered: the fp32 tile kernel with that selection mode; prefilter: the bf16 prefilter + exact re-rank kernel
    wherever it applies (un-split, normalised problems)."""
    if select == "prefilter":
        monkeypatch.delenv("GKG_KNN_SELECT", raising=False)
        monkeypatch.setenv("GKG_KNN_PREFILTER", "force")
    else:
        monkeypatch.setenv("GKG_KNN_SELECT", select)


@pytest.mark.parametrize("select", ["direct", "buffered", "prefilter"])
def test_fuzz_random_shapes_bit_exact(select, monkeypatch):
    _set_mode(monkeypatch, select)
    _fuzz_random_shapes()


def _fuzz_random_shapes():
    """60 random (BG, c, N, M, k, d, relpos, dtype) problems incl. ragged sizes (c not a multiple of 8, N/M not multiples
    of the 64/32 tiles, k*d up to 64, self and bipartite graphs): graph, aggregation and argmax bit-exact vs the C
    oracle, backward within rounding."""
    from gkgnet_amd import ops
    from oracle import c_oracle as O
    rng = np.random.RandomState(2024)
    for it in range(60):
        BG = int(rng.randint(1, 5))
        c = int(rng.choice([1, 3, 4, 7, 8, 12, 20, 33, 48, 64, 100]))
        N = int(rng.randint(1, 200))
        self_graph = rng.rand() < 0.4
        M = N if self_graph else int(rng.randint(1, 300))
        kd_max = min(M, 64)
        d = int(rng.randint(1, 4))
        k = int(rng.randint(1, max(2, kd_max // d + 1)))
        if k * d > kd_max:
            d = 1
            k = min(k, kd_max)
        use_rp = rng.rand() < 0.5
        bf16 = rng.rand() < 0.25
        x = rng.standard_normal((BG, c, N)).astype(np.float32)
        y = None if self_graph else rng.standard_normal((BG, c, M)).astype(np.float32)
        rp = -rng.random_sample((N, M)).astype(np.float32) if use_rp else None
        tdt = torch.bfloat16 if bf16 else torch.float32
        xd = torch.from_numpy(x).to(tdt)
        yd = None if y is None else torch.from_numpy(y).to(tdt)
        xo = xd.float().numpy()
        yo = None if yd is None else yd.float().numpy()
        tag = (it, BG, c, N, M, k, d, use_rp, bf16)
        want_idx, want_center = O.knn(xo, yo, rp, k, d)
        edge = ops.knn_graph(xd.cuda(), None if yd is None else yd.cuda(), None if rp is None else _dev(rp).unsqueeze(0), k, d)
        assert np.array_equal(edge[0].cpu().numpy(), want_idx), tag
        assert np.array_equal(edge[1].cpu().numpy(), want_center), tag
        want_m, want_arg = O.mr_fwd(xo, yo, want_idx)
        xg = xd.cuda().requires_grad_(True)
        yg = None if yd is None else yd.cuda().requires_grad_(True)
        m = ops.max_relative(xg, edge[0], yg)
        ref_m = torch.from_numpy(want_m).to(tdt).float().numpy()
        assert np.array_equal(m.detach().float().cpu().numpy(), ref_m), tag
        if not bf16:
            g = rng.standard_normal(want_m.shape).astype(np.float32)
            m.backward(_dev(g))
            gx, gs = O.mr_bwd(g, want_idx, want_arg, None if y is None else M)
            assert np.allclose(xg.grad.cpu().numpy(), gx, atol=1e-5, rtol=1e-5), tag
            if y is not None:
                assert np.allclose(yg.grad.cpu().numpy(), gs, atol=1e-5, rtol=1e-5), tag


def test_non_finite_inputs_do_not_crash_or_leave_the_index_range():
    """NaN / Inf features give unspecified neighbours (the reference's topk order with NaN is unspecified too), but the
    operators must stay in range and must not fault: indices in [0, M), aggregation finite where the inputs are."""
    from gkgnet_amd import ops
    torch.manual_seed(3)
    for N, M in ((50, None), (20, 3000)):                # single-workgroup path and split-key + merge path
        x = torch.randn(2, 8, N, device="cuda")
        y = None if M is None else torch.randn(2, 8, M, device="cuda")
        x[0, :, 3] = float("nan")
        x[1, 2, 7] = float("inf")
        if y is not None:
            y[0, :, 5] = float("nan")
        e = ops.knn_graph(x, y, None, 9, 1)
        Mk = N if M is None else M
        assert int(e[0].min()) >= 0 and int(e[0].max()) < Mk
        m = ops.max_relative(x, e[0], y)
        torch.cuda.synchronize()
        assert m.shape == x.shape
    # corrupted index tensors are clamped instead of read out of bounds
    x = torch.randn(1, 4, 10, device="cuda", requires_grad=True)
    bad = torch.full((1, 10, 3), 10**9, dtype=torch.int64, device="cuda")
    bad[0, :, 1] = -5
    m = ops.max_relative(x, bad)
    m.sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(m).all() and torch.isfinite(x.grad).all()


@pytest.mark.parametrize("select", ["direct", "buffered", "prefilter"])
def test_exact_ties_decide_membership_and_order(select, monkeypatch):
    _set_mode(monkeypatch, select)
    _exact_ties()


def _exact_ties():
    """Inputs built to tie EXACTLY (keys duplicated three times; small-integer features without normalisation) for long
    and short lists: 'equal distance -> smaller key index first' must decide both which neighbours enter the top
    k*d and their order (hence which survive the dilation), bit for bit like the C oracle."""
    from gkgnet_amd import ops
    from oracle import c_oracle as O
    rng = np.random.RandomState(77)
    # (BG, c, N, M, self, k, d, relpos, kind)
    cases = [(2, 12, 90, 150, False, 9, 2, False, "dup"), (1, 16, 130, 130, True, 12, 2, True, "dup"),
             (2, 40, 200, 1200, False, 9, 1, False, "dup"), (3, 20, 70, 70, True, 3, 3, True, "dup"),
             (2, 8, 70, 200, False, 6, 3, False, "int"), (1, 4, 64, 64, True, 18, 1, False, "int"),
             (1, 3, 33, 500, False, 32, 2, False, "int")]
    for BG, c, N, M, self_graph, k, d, use_rp, kind in cases:
        normalize = kind != "int"
        if kind == "int":
            x = rng.randint(-2, 3, size=(BG, c, N)).astype(np.float32)
            y = None if self_graph else rng.randint(-2, 3, size=(BG, c, M)).astype(np.float32)
        else:
            x = rng.standard_normal((BG, c, N)).astype(np.float32)
            y = None if self_graph else rng.standard_normal((BG, c, M)).astype(np.float32)
            t = x if self_graph else y
            T = t.shape[2] // 3
            t[:, :, T:2 * T] = t[:, :, :T]
            t[:, :, 2 * T:3 * T] = t[:, :, :T]
        rp = (np.round(-rng.random_sample((N, M)) * 4) / 4).astype(np.float32) if use_rp else None   # coarse: ties survive
        tag = (BG, c, N, M, self_graph, k, d, use_rp, kind)
        want_idx, want_center = O.knn(x, y, rp, k, d, normalize=normalize)
        got = ops.knn_graph(_dev(x), _dev(y), None if rp is None else _dev(rp).unsqueeze(0), k, d, normalize).cpu().numpy()
        assert np.array_equal(got[0], want_idx), tag
        assert np.array_equal(got[1], want_center), tag


def test_prefilter_slow_path_on_massive_ties(monkeypatch):
    """The prefilter kernel's slow path (a query with more survivors than it can re-rank, or a wave whose list may have
    dropped one, re-scans all keys with the exact chain): key sets with far more exact ties than a list holds — every key
    identical; 40 copies of each of 5 keys — must still come out in the contract's order (smaller key index first), bit
    for bit like the C oracle."""
    from gkgnet_amd import ops
    from oracle import c_oracle as O
    monkeypatch.delenv("GKG_KNN_SELECT", raising=False)
    monkeypatch.setenv("GKG_KNN_PREFILTER", "force")
    rng = np.random.RandomState(5)
    for c, N, M, k, d, nuniq in ((32, 70, 200, 9, 1, 1), (48, 100, 200, 9, 2, 5), (16, 65, 130, 12, 1, 2), (64, 40, 1300, 9, 3, 4)):
        base = rng.standard_normal((2, c, nuniq)).astype(np.float32)
        y = np.ascontiguousarray(base[:, :, rng.randint(0, nuniq, size=M)])
        x = rng.standard_normal((2, c, N)).astype(np.float32)
        want_idx, _ = O.knn(x, y, None, k, d)
        edge = ops.knn_graph(_dev(x), _dev(y), None, k, d)
        assert np.array_equal(edge[0].cpu().numpy(), want_idx), (c, N, M, k, d, nuniq)


def test_prefilter_is_not_taken_for_a_bias_outside_its_error_bound(monkeypatch):
    """the prefilter's error bound assumes |relative_pos| <= 1.  With a bias far outside that range (|rp| up to
    300, where the matrix-core accumulation error exceeds the margin) the call must still return the contract's graph even
    with the prefilter forced: the Python layer does not vouch for the range (no GKG_KNN_RELPOS_UNIT) and the library
    takes the fp32 tile kernel."""
    from gkgnet_amd import _lib, ops
    from oracle import c_oracle as O
    monkeypatch.delenv("GKG_KNN_SELECT", raising=False)
    monkeypatch.setenv("GKG_KNN_PREFILTER", "force")
    rng = np.random.RandomState(11)
    x = rng.standard_normal((2, 64, 150)).astype(np.float32)
    y = rng.standard_normal((2, 64, 1100)).astype(np.float32)
    # near-equal large biases: candidates separated by the fp32 rounding of (bias + distance), not by the bias itself
    rp = (-300.0 + 1e-3 * rng.random_sample((150, 1100))).astype(np.float32)
    rp_dev = _dev(rp).unsqueeze(0)
    assert _lib.relpos_flags(rp_dev) == 0
    small = _dev(-rng.random_sample((150, 1100)).astype(np.float32)).unsqueeze(0)
    assert _lib.relpos_flags(small) == _lib.KNN_RELPOS_UNIT
    want_idx, _ = O.knn(x, y, rp, 9, 2)
    edge = ops.knn_graph(_dev(x), _dev(y), rp_dev, 9, 2)
    assert np.array_equal(edge[0].cpu().numpy(), want_idx)
