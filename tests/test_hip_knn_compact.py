"""Compact graph inside the block (round 5): gkg_knn_fwd_tm16 writes the neighbour lists as u16 rows instead of the
(2, B*G, N, k) int64 edge_index, gkg_mr_fwd_tm16 / gkg_mr_linear_bf16_nn16 read them (reference chain torch_edge.py:164-176 ->
torch_vertex.py:49-61; Grapher.forward discards the graph, torch_vertex.py:330).  Same neighbours in the same order, same
aggregation bits as the int64 path, at every real layer shape and in every k-NN kernel mode; the fused block takes the compact
path by itself and its outputs / gradients equal the int64 path's."""
import os

import pytest
import torch

from test_hip_config_shapes import SHAPES, GROUPS, _case, _dev, knn_select      # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _tm(a):
    """(BG, c, T) channel-major problem stack -> (B = 1, T, C = BG * c) token-major with G = BG groups."""
    BG, c, T = a.shape
    return a.permute(2, 0, 1).reshape(1, T, BG * c).contiguous()


@pytest.mark.parametrize("name", sorted(SHAPES))
def test_compact_lists_equal_the_int64_graph(name, knn_select):
    from gkgnet_amd import fused
    c, N, M, k, d, relpos = SHAPES[name]
    if (M or N) > 65536:
        pytest.skip("u16 rows")
    BG = 2
    x, y, rp, k, d = _case(name, BG)
    xt, yt = _tm(_dev(x)), (None if y is None else _tm(_dev(y)))
    rpt = None if rp is None else _dev(rp).unsqueeze(0)
    edge = fused.knn_graph_tm(xt, yt, rpt, k, d, BG)
    nn16 = fused.knn_graph_tm16(xt, yt, rpt, k, d, BG)
    assert nn16.dtype == torch.int16 and tuple(nn16.shape) == (BG, N, k)
    assert torch.equal(nn16.to(torch.int64) & 0xFFFF, edge[0])


@pytest.mark.parametrize("name", ["s4_grapher_d3", "s3_grapher_d2", "s2_grapher_r2", "m4_grapher_d2", "s3_label"])
@pytest.mark.parametrize("mode", [0, 1])
def test_aggregation_over_compact_lists_is_bit_identical(name, mode):
    from gkgnet_amd import fused
    c, N, M, k, d, relpos = SHAPES[name]
    BG = 4
    if mode == 1 and (BG * c) % 16:
        pytest.skip("mode 1 needs C % 16 == 0")
    x, y, rp, k, d = _case(name, BG)
    xt, yt = _tm(_dev(x)), (None if y is None else _tm(_dev(y)))
    rpt = None if rp is None else _dev(rp).unsqueeze(0)
    edge = fused.knn_graph_tm(xt, yt, rpt, k, d, BG)
    nn16 = fused.knn_graph_tm16(xt, yt, rpt, k, d, BG)
    outs = []
    for nn in (edge[0], nn16):
        xg = xt.clone().requires_grad_(True)
        sg = None if yt is None else yt.clone().requires_grad_(True)
        o = fused._MaxRelativeTM.apply(xg, sg, nn, BG, mode)
        g = torch.randn(o.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
        o.backward(g)
        outs.append((o.detach(), xg.grad, None if sg is None else sg.grad))
    assert torch.equal(outs[0][0], outs[1][0])
    # the backward scatters from the saved winning rows in both cases (same inputs); where it is the fp32-atomic form (few query
    # rows over > 512 keys) its own summation order varies from run to run
    exact = not (N < 160 and (M or N) > 512)
    same = torch.equal if exact else (lambda a, b: torch.allclose(a, b, atol=1e-5, rtol=1e-5))
    assert same(outs[0][1], outs[1][1])
    assert outs[0][2] is None or same(outs[0][2], outs[1][2])


def _grapher(C, G, k, HW, r, seed):
    from gkgnet_amd.grapher import Grapher
    torch.manual_seed(seed)
    m = Grapher(C, kernel_size=k, dilation=1, conv="mr", act="gelu", norm="batch", bias=True, stochastic=False, epsilon=0.2,
                r=r, n=HW * HW, drop_path=0.0, relative_pos=True, use_multi_group=True, num_group=G).cuda()
    return m


@pytest.mark.parametrize("C,G,HW,r,train", [(64, 2, 24, 2, True), (64, 2, 24, 2, False), (96, 4, 36, 1, True)])
def test_block_takes_the_compact_path_and_matches_the_int64_path(C, G, HW, r, train, monkeypatch):
    """A Grapher whose graph does not take the one-kernel form (key counts above the fused form's lists / pooled keys through the
    prefilter ...) runs k-NN -> aggregation over u16 lists; forcing the int64 path (GKG_DISABLE-style switch) gives the same
    output bit for bit and the same gradients."""
    from gkgnet_amd import block, fused
    monkeypatch.setattr(block, "ENABLED", False)             # the per-layer composition's calls are counted below
    m = _grapher(C, G, 9, HW, r, 3)
    m.train(train)
    x = torch.randn(2, C, HW, HW, device="cuda", generator=torch.Generator(device="cuda").manual_seed(4))
    calls = {"c16": 0}
    real16 = fused.knn_graph_tm16

    def spy(*a, **kw):
        calls["c16"] += 1
        return real16(*a, **kw)
    monkeypatch.setattr(fused, "knn_graph_tm16", spy)
    monkeypatch.setattr(fused, "KNN_MR", False)              # the two-launch form (the one-kernel form has no index tensor at all)
    res = []
    for compact in (True, False):
        monkeypatch.setattr(fused, "KNN_COMPACT", compact)
        xg = x.clone().requires_grad_(train)
        with torch.set_grad_enabled(train):
            out = m(xg)
        if train:
            m.zero_grad(set_to_none=True)
            out.backward(torch.ones_like(out))
            res.append((out.detach(), xg.grad.clone(), m.fc1[0].weight.grad.clone()))
        else:
            res.append((out.detach(),))
    assert calls["c16"] == 1
    assert torch.equal(res[0][0], res[1][0])                     # same graph, same aggregation: the same forward bits
    for a, b in zip(res[0][1:], res[1][1:]):                    # gradients: the step's atomic sums (weight gradients, BN backward)
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-12      # vary in the last bits from run to run


def test_fuzz_compact_lists_random_shapes():
    """Random (B, G, c, N, M, k, dilation, bias, graph kind): the u16 lists equal the int64 plane for every launch plan the sizes
    select (key splits with the merge kernel, single-wave workgroups, buffered / direct selection, prefilter)."""
    import numpy as np
    from gkgnet_amd import fused
    rng = np.random.RandomState(5)
    gen = torch.Generator(device="cuda").manual_seed(5)
    for trial in range(40):
        G = int(rng.choice([1, 2, 4, 8]))
        c = 4 * int(rng.randint(1, 26))
        B = int(rng.randint(1, 5))
        N = int(rng.choice([rng.randint(1, 100), rng.randint(100, 600), rng.randint(600, 2500)]))
        self_graph = bool(rng.rand() < 0.5)
        M = N if self_graph else int(rng.choice([rng.randint(20, 300), rng.randint(300, 3000)]))
        d = int(rng.randint(1, 4))
        k = int(rng.randint(1, 19))
        if k * d > min(M, 64):
            d, k = 1, min(k, M, 64)
        x = torch.randn(B, N, G * c, device="cuda", generator=gen)
        y = None if self_graph else torch.randn(B, M, G * c, device="cuda", generator=gen)
        rp = -torch.rand(1, N, M, device="cuda", generator=gen) if rng.rand() < 0.5 else None
        edge = fused.knn_graph_tm(x, y, rp, k, d, G)
        nn16 = fused.knn_graph_tm16(x, y, rp, k, d, G)
        assert torch.equal(nn16.to(torch.int64) & 0xFFFF, edge[0]), (trial, B, G, c, N, M, k, d)
