"""GPU parity at the layer shapes of BASELINE configs 3/4 (GKGNet-576, pvig_s) and 5 (pvig_m @ 768, k=18, G=8).

(a) Operator level, BG = 2 problems per shape: graph, aggregation and argmax bit-exact against the C oracle, backward
    within rounding — through BOTH C-ABI layouts (channel-major gkg_knn_fwd / gkg_mr_*, token-major *_tm used by the
    fused block).  Shapes: SURVEY.md §8 per-layer table (reference gkgnet.py:180-183,234: reduce_ratios [4,2,1,1],
    dilation min(idx//4+1, 49//k)) and its pvig_m counterpart (blocks [2,2,16,2], channels [96,192,384,768]).
(b) Full batch of the config (B = 32 / 16): three of the launch's problems bit-exact against the C oracle, and
    size-independent properties of ALL of them checked against a dense fp64 evaluation on the GPU in problem chunks — in-range distinct neighbours, ascending distances, top-(k*d) optimality, dilation
    picks ranks 0,d,2d.., aggregation equal to the dense gather/max bit for bit, gradient mass conservation.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# name: (c, N, M(None = self graph), k, d, relpos)          BG = 2 in the bit-exact tests
PVIG_S_576 = {
    "s1_grapher_r4": (40, 20736, 1296, 9, 1, True),
    "s1_label": (40, 80, 20736, 9, 1, False),
    "s2_grapher_r2": (80, 5184, 1296, 9, 1, True),
    "s2_label": (80, 80, 5184, 9, 1, False),
    "s3_grapher_d2": (200, 1296, None, 9, 2, True),
    "s3_grapher_d3": (200, 1296, None, 9, 3, True),
    "s3_label": (200, 80, 1296, 9, 1, False),
    "s4_grapher_d3": (320, 324, None, 9, 3, True),
    "s4_label": (320, 80, 324, 9, 1, False),
}
PVIG_M_768 = {
    "m1_grapher_r4": (12, 36864, 2304, 18, 1, True),
    "m1_label": (12, 80, 36864, 18, 1, False),
    "m2_grapher_r2": (24, 9216, 2304, 18, 1, True),
    "m2_label": (24, 80, 9216, 18, 1, False),
    "m3_grapher_d2": (48, 2304, None, 18, 2, True),
    "m3_label": (48, 80, 2304, 18, 1, False),
    "m4_grapher_d2": (96, 576, None, 18, 2, True),
    "m4_label": (96, 80, 576, 18, 1, False),
}
SHAPES = {**PVIG_S_576, **PVIG_M_768}
GROUPS = {**{k: 2 for k in PVIG_S_576}, **{k: 8 for k in PVIG_M_768}}


def _seed(name):
    import zlib
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def _case(name, BG):
    c, N, M, k, d, relpos = SHAPES[name]
    rng = np.random.RandomState(_seed(name))
    x = rng.standard_normal((BG, c, N)).astype(np.float32)
    y = None if M is None else rng.standard_normal((BG, c, M)).astype(np.float32)
    rp = -rng.random_sample((N, N if M is None else M)).astype(np.float32) if relpos else None
    return x, y, rp, k, d


def _dev(a, dtype=torch.float32):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to("cuda", dtype)


@pytest.fixture(params=["auto", "direct", "buffered", "prefilter"])
def knn_select(request):
    """The k-NN kernel's mode: the library's own rule, one selection mode of the fp32 tile kernel forced, or the bf16
    prefilter + exact re-rank kernel forced wherever it applies (GKG_KNN_SELECT / GKG_KNN_PREFILTER are read per call)."""
    import os
    old = {k: os.environ.get(k) for k in ("GKG_KNN_SELECT", "GKG_KNN_PREFILTER")}
    os.environ.pop("GKG_KNN_SELECT", None)
    os.environ.pop("GKG_KNN_PREFILTER", None)
    if request.param == "prefilter":
        os.environ["GKG_KNN_PREFILTER"] = "force"
    elif request.param != "auto":
        os.environ["GKG_KNN_SELECT"] = request.param
    yield request.param
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_layer_shape_bit_exact_vs_c_oracle(name, knn_select):
    from gkgnet_amd import fused, ops
    from oracle import c_oracle as O
    BG = 2
    x, y, rp, k, d = _case(name, BG)
    c, N = x.shape[1:]
    M = None if y is None else y.shape[2]
    want_idx, want_center = O.knn(x, y, rp, k, d)
    # ---- channel-major C-ABI (the reference's (B*G, c, N) layout)
    edge = ops.knn_graph(_dev(x), _dev(y), None if rp is None else _dev(rp).unsqueeze(0), k, d)
    got = edge.cpu().numpy()
    assert np.array_equal(got[0], want_idx), name
    assert np.array_equal(got[1], want_center), name
    want_m, want_arg = O.mr_fwd(x, y, want_idx)
    xd = _dev(x).requires_grad_(True)
    yd = None if y is None else _dev(y).requires_grad_(True)
    m = ops.max_relative(xd, edge[0], yd)
    assert np.array_equal(m.detach().cpu().numpy(), want_m), name
    g = np.random.RandomState(5).standard_normal(want_m.shape).astype(np.float32)
    m.backward(_dev(g))
    want_gx, want_gsrc = O.mr_bwd(g, want_idx, want_arg, M)
    assert np.allclose(xd.grad.cpu().numpy(), want_gx, atol=2e-5, rtol=1e-5), name
    if y is not None:
        assert np.allclose(yd.grad.cpu().numpy(), want_gsrc, atol=2e-5, rtol=1e-5), name
    # ---- token-major C-ABI (what the fused block calls): the BG = 2 problems are the G = 2 groups of ONE image
    if c % 4 == 0:
        G = BG
        xtm = _dev(x).permute(2, 0, 1).reshape(1, N, G * c).contiguous()            # (B=1, N, C)
        ytm = None if y is None else _dev(y).permute(2, 0, 1).reshape(1, -1, G * c).contiguous()
        e_tm = fused.knn_graph_tm(xtm, ytm, None if rp is None else _dev(rp).unsqueeze(0), k, d, G)
        assert torch.equal(e_tm, edge), name
        xg = xtm.clone().requires_grad_(True)
        yg = None if ytm is None else ytm.clone().requires_grad_(True)
        m_tm = fused._MaxRelativeTM.apply(xg, yg, e_tm[0], G, 0)
        assert np.array_equal(m_tm.detach().reshape(N, G, c).permute(1, 2, 0).cpu().numpy(), want_m), name
        m_tm.backward(_dev(g).permute(2, 0, 1).reshape(1, N, G * c).contiguous())
        assert np.allclose(xg.grad.reshape(N, G, c).permute(1, 2, 0).cpu().numpy(), want_gx, atol=2e-5, rtol=1e-5), name
        if y is not None:
            assert np.allclose(yg.grad.reshape(-1, G, c).permute(1, 2, 0).cpu().numpy(), want_gsrc, atol=2e-5,
                               rtol=1e-5), name


# (name, full batch of the config)
FULL = [("s1_grapher_r4", 32), ("s1_label", 32), ("s2_grapher_r2", 32), ("s3_grapher_d2", 32), ("s3_grapher_d3", 32),
        ("s4_grapher_d3", 32), ("m1_grapher_r4", 16), ("m1_label", 16), ("m2_grapher_r2", 16), ("m3_grapher_d2", 16),
        ("m4_grapher_d2", 16)]


@pytest.mark.parametrize("name,B", FULL)
def test_full_batch_properties(name, B):
    """The config's full batch (BG = B*G problems) through the HIP operators; properties against dense fp64 math on
    the GPU, evaluated in chunks of problems so the (chunk, N, M) fp64 matrix stays below ~2 GB."""
    from gkgnet_amd import ops
    c, N, M, k, d, relpos = SHAPES[name]
    BG = B * GROUPS[name]
    Mk = N if M is None else M
    gen = torch.Generator(device="cuda").manual_seed(_seed(name))
    x = torch.randn(BG, c, N, device="cuda", generator=gen)
    y = None if M is None else torch.randn(BG, c, M, device="cuda", generator=gen)
    rp = -torch.rand(1, N, Mk, device="cuda", generator=gen) if relpos else None
    edge = ops.knn_graph(x, y, rp, k, d)
    nn_idx = edge[0]
    assert nn_idx.shape == (BG, N, k) and int(nn_idx.min()) >= 0 and int(nn_idx.max()) < Mk
    srt = nn_idx.sort(dim=-1).values
    assert bool((srt[..., 1:] != srt[..., :-1]).all()), "duplicate neighbours"
    assert torch.equal(edge[1], torch.arange(N, device="cuda").view(1, N, 1).expand(BG, N, k))
    # bit-exact against the C oracle on a SAMPLE of the launch's problems (first, a middle one, last): the full-batch launch is
    # where the interleaved XCD map, the single-batch prefilter form and the single-wave forms engage
    from oracle import c_oracle as O
    rp_np = None if rp is None else rp[0].cpu().numpy()
    for bg in sorted({0, BG // 2 + 1, BG - 1}):
        want_idx, _ = O.knn(x[bg:bg + 1].cpu().numpy(), None if y is None else y[bg:bg + 1].cpu().numpy(), rp_np, k, d)
        assert np.array_equal(nn_idx[bg:bg + 1].cpu().numpy(), want_idx), (name, bg)
    kd = k * d
    chunk = max(1, int(2e9 // (8 * N * Mk)))
    src = x if y is None else y
    for b0 in range(0, BG, chunk):
        sl = slice(b0, min(BG, b0 + chunk))
        xn = torch.nn.functional.normalize(x[sl].double(), dim=1)
        yn = xn if y is None else torch.nn.functional.normalize(y[sl].double(), dim=1)
        dist = (xn * xn).sum(1).unsqueeze(-1) - 2 * xn.transpose(1, 2) @ yn + (yn * yn).sum(1).unsqueeze(1)
        if rp is not None:
            dist = dist + rp.double()
        dsel = torch.gather(dist, 2, nn_idx[sl])
        # ascending kept ranks; rank j*d has exactly j*d strictly-better keys (up to fp32 near-ties, 2e-6)
        assert (dsel[..., 1:] - dsel[..., :-1]).min().item() > -2e-6, name
        for j in (0, k - 1):
            nb = (dist < dsel[..., j:j + 1] - 2e-6).sum(-1)
            nw = (dist <= dsel[..., j:j + 1] + 2e-6).sum(-1)
            assert bool((nb <= j * d).all()) and bool((nw >= j * d + 1).all()), (name, j)
        del dist, dsel
    # aggregation: bit-exact against the dense gather/max, chunked over problems
    m = ops.max_relative(x, nn_idx, y)
    cb = max(1, int(1e9 // (4 * c * N * k)))
    for b0 in range(0, BG, cb):
        sl = slice(b0, min(BG, b0 + cb))
        idx = nn_idx[sl].reshape(sl.stop - sl.start, 1, N * k).expand(-1, c, -1)
        want = (torch.gather(src[sl], 2, idx).view(-1, c, N, k) - x[sl].unsqueeze(-1)).max(-1).values
        assert torch.equal(m[sl], want), name
    # backward: every g is subtracted once (centre) and added once (argmax neighbour)
    xr = x.clone().requires_grad_(True)
    yr = None if y is None else y.clone().requires_grad_(True)
    g = torch.randn(BG, c, N, device="cuda", generator=gen)
    ops.max_relative(xr, nn_idx, yr).backward(g)
    tot = xr.grad.double().sum(-1) + (0 if yr is None else yr.grad.double().sum(-1))
    assert tot.abs().max().item() < 1e-3 * max(1.0, float(N) ** 0.5), name
    if yr is not None:
        assert torch.allclose(xr.grad, -g)
