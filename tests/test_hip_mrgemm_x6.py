"""SURVEY §8 row g1, TRAINING / fp32 form (csrc/gkg_mrgemm_x6.hip): gather + max(x_j - x_i) + interleave + grouped 1x1
projection (+ BN column sums) in ONE launch, checked against the C ORACLE's aggregation (oracle/c_oracle.mr_fwd: bit-exact m
and argmax) followed by a dense grouped convolution of the oracle's interleaved output (fp64 F.conv2d, groups = 4), then
forward + backward of the autograd node against the un-fused launches and against the oracle's module (tolerance 1e-3,
north_star).  Reference: torch_vertex.py:47-62, torch_nn.py:57-69."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (B, G, C, N, M (None = self graph), k)
CASES = [(2, 4, 320, 324, None, 9), (4, 4, 320, 80, 324, 9), (2, 2, 80, 150, None, 9), (3, 2, 160, 100, 25, 9),
         (1, 2, 400, 300, None, 9), (2, 2, 640, 70, None, 9), (2, 8, 192, 333, 90, 18), (2, 1, 48, 65, None, 5),
         (1, 2, 16, 64, None, 3), (2, 3, 288, 100, None, 9), (1, 8, 768, 40, None, 4)]


def test_supported_query_matches_the_lds_budget():
    from gkgnet_amd import _lib
    lib = _lib.load()
    assert lib.gkg_mr_linear_x6_supported(4, 80, 9) == 1 and lib.gkg_mr_linear_x6_supported(2, 320, 9) == 1
    assert lib.gkg_mr_linear_x6_supported(8, 96, 4) == 1
    assert lib.gkg_mr_linear_x6_supported(8, 96, 18) == 0          # pvig_m stage 4, k = 18: the index rows no longer fit
    assert lib.gkg_mr_linear_x6_supported(2, 5, 9) == 0            # C % 16



def _oracle(x, src, idx, G, weight):
    """(m, row-argmax, u, y) from the oracle: m / argmax = oracle mr_fwd on the reference's (B*G, c, N) layout; u = the
    reference's interleave (torch_vertex.py:61); y = F.conv2d(u, W, groups=4) evaluated in fp64."""
    from oracle import c_oracle as O
    B, N, C = x.shape
    cg = C // G
    xc = x.permute(0, 2, 1).reshape(B * G, cg, N).cpu().numpy()
    sc = None if src is None else src.permute(0, 2, 1).reshape(B * G, cg, src.shape[1]).cpu().numpy()
    idn = idx.cpu().numpy()
    m, slot = O.mr_fwd(xc, sc, idn)                                   # (BG, c, N) each
    rows = np.take_along_axis(idn[:, None, :, :].repeat(cg, 1), slot[..., None].astype(np.int64), axis=3)[..., 0]   # (BG, c, N)
    m_t = torch.from_numpy(m).reshape(B, C, N)
    rows_t = torch.from_numpy(rows).reshape(B, C, N).permute(0, 2, 1).contiguous()                      # (B, N, C)
    xr = x.permute(0, 2, 1).cpu()                                                                       # (B, C, N)
    u = torch.cat([xr.unsqueeze(2), m_t.unsqueeze(2)], dim=2).reshape(B, 2 * C, N, 1)                   # torch_vertex.py:61
    y = torch.nn.functional.conv2d(u.double(), weight.detach().cpu().double(), None, groups=4)          # (B, 2C, N, 1)
    return m_t, rows_t, u, y.reshape(B, 2 * C, N)


@pytest.mark.parametrize("save_u", [True, False])
@pytest.mark.parametrize("B,G,C,N,M,k", CASES)
def test_fused_kernel_against_oracle_aggregation_and_dense_conv(B, G, C, N, M, k, save_u):
    from gkgnet_amd import _lib, fused
    from gkgnet_amd.ops import _ptr, _stream
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(C * 7 + N)
    x = torch.randn(B, N, C, device="cuda", generator=gen)
    src = None if M is None else torch.randn(B, M, C, device="cuda", generator=gen)
    Mk = N if M is None else M
    idx = torch.randint(0, Mk, (B * G, N, k), device="cuda", generator=gen)
    conv = torch.nn.Conv2d(2 * C, 2 * C, 1, groups=4).cuda()
    T, ci, co = B * N, C // 2, C // 2
    pf, _ = fused._planes(lib, conv.weight, 4, co, ci, True, False)
    y = torch.full((4, T, co), float("nan"), device="cuda")
    arg = torch.zeros((B, N, C), dtype=torch.int16, device="cuda")
    u = torch.full((4, T, ci), float("nan"), device="cuda") if save_u else None
    sums = torch.zeros(4 * 2 * co, dtype=torch.float64, device="cuda")
    _lib.check(lib.gkg_mr_linear_x6(_ptr(x), _ptr(src), _ptr(idx), _ptr(pf), _ptr(y), _ptr(arg), _ptr(u), _ptr(sums),
                                    B, G, C // G, N, Mk, k, _stream()), "gkg_mr_linear_x6")
    torch.cuda.synchronize()
    m_o, rows_o, u_o, y_o = _oracle(x, src, idx, G, conv.weight)
    # winning rows: bit-exact (ties resolve to the first maximum on both sides)
    got_rows = (arg.cpu().numpy().astype(np.int32) & 0xffff)
    assert np.array_equal(got_rows, rows_o.numpy()), float((got_rows != rows_o.numpy()).mean())
    # y: conv group q's rows are output channels [q co, (q + 1) co) of the reference's BasicConv output (bias excluded)
    y_got = y.view(4, B, N, co).permute(1, 0, 3, 2).reshape(B, 2 * C, N).double().cpu()
    scale = float(y_o.abs().max())
    err = float((y_got - y_o).abs().max())
    assert err <= 2e-5 * max(scale, 1.0), (err, scale)                  # x6 arithmetic: below an fp32 fma chain's error
    # column sums for the batch statistics: sum y and sum y^2 over all tokens, per output channel
    s_want = y_got.permute(1, 0, 2).reshape(2 * C, -1)
    got = sums.view(4, 2, co).cpu()
    assert torch.allclose(got[:, 0].reshape(-1), s_want.sum(1), rtol=1e-9, atol=1e-6 * max(scale, 1.0) * T)
    assert torch.allclose(got[:, 1].reshape(-1), (s_want * s_want).sum(1), rtol=1e-6, atol=1e-9)
    # the interleaved operand: bit-identical to the oracle's [x, m] whether stored by the forward or rebuilt from the rows
    u_want = u_o.reshape(B, 4, ci, N).permute(1, 0, 3, 2).reshape(4, T, ci)
    if not save_u:
        u = torch.full((4, T, ci), float("nan"), device="cuda")
        _lib.check(lib.gkg_mr_regather_tm(_ptr(x), _ptr(src), _ptr(arg), _ptr(u), B, N, Mk, C, _stream()), "gkg_mr_regather_tm")
    assert torch.equal(u.cpu(), u_want)


@pytest.mark.parametrize("save_u", [True, False])
@pytest.mark.parametrize("label", [False, True])
def test_autograd_node_matches_the_separate_launches(label, save_u, monkeypatch):
    """Forward + backward of the fused node == aggregation kernel + grouped projection + BN passes on identical inputs
    (the un-fused path is pinned to the reference's fixtures F1-F7): outputs, input / key gradients, weight and BN
    gradients, running statistics."""
    from gkgnet_amd import fused, layers
    layers.norm_cfg["type"] = "BN"
    monkeypatch.setattr(fused, "MR_SAVE_U", save_u)
    B, G, C, N, k = 4, 4, 320, 80 if label else 324, 9
    M = 324 if label else N
    gen = torch.Generator(device="cuda").manual_seed(5)
    x0 = torch.randn(B, N, C, device="cuda", generator=gen)
    s0 = torch.randn(B, M, C, device="cuda", generator=gen) if label else None
    idx = torch.randint(0, M, (B * G, N, k), device="cuda", generator=gen)
    cot = torch.randn(B * N, 2 * C, device="cuda", generator=gen)
    res = []
    for fusedk in (True, False):
        torch.manual_seed(3)
        conv = torch.nn.Conv2d(2 * C, 2 * C, 1, groups=4).cuda()
        bn = layers.build_norm(2 * C).cuda().train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.2)
        nn_ = torch.nn.Sequential(conv, bn, torch.nn.GELU())
        monkeypatch.setattr(fused, "MR_X6", fusedk)
        x = x0.clone().requires_grad_(True)
        s = None if s0 is None else s0.clone().requires_grad_(True)
        assert fused._mr_x6_ok(x, s, nn_, C, G, k) == fusedk
        out = fused._aggregate_project(x, s, idx, G, nn_, C, False)
        out.backward(cot)
        res.append(dict(out=out.detach(), gx=x.grad, gs=None if s is None else s.grad, dw=conv.weight.grad, dg=bn.weight.grad,
                        db=bn.bias.grad, rm=bn.running_mean.clone(), rv=bn.running_var.clone()))
    a, b = res
    for key in a:
        if a[key] is None:
            assert b[key] is None
            continue
        ref = b[key]
        tol = 1e-4 * max(1.0, float(ref.abs().max()))
        assert float((a[key] - ref).abs().max()) <= tol, (key, float((a[key] - ref).abs().max()), tol)


def test_blocks_take_the_fused_launch_in_an_fp32_train_step(monkeypatch):
    """Grapher + GrapherLabel, fp32, train mode, GKG_ENABLE=mr_x6: MRConv2d.forward runs as the fused launch (no stand-alone
    gkg_mr_fwd_tm), and the step matches the torch-CPU oracle within the 1e-3 contract (forward and input gradients)."""
    from gkgnet_amd import _lib, fused, layers
    monkeypatch.setattr(fused, "MR_X6", True)
    from gkgnet_amd.grapher import Grapher, GrapherLabel
    from oracle import torch_ref as R
    layers.norm_cfg["type"] = "BN"
    torch.manual_seed(0)
    C, G, H, L, B = 64, 4, 10, 12, 3
    g = Grapher(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, drop_path=0.0, relative_pos=True,
                use_multi_group=True, num_group=G).cuda().train()
    gl = GrapherLabel(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, drop_path=0.0, relative_pos=False,
                      num_nodes=L, use_multi_group=True, num_group=G).cuda().train()
    pg = {k: v.detach().cpu().clone() for k, v in g.state_dict().items()}
    pl = {k: v.detach().cpu().clone() for k, v in gl.state_dict().items()}
    xin, ein = torch.randn(B, C, H, H), torch.randn(B, L, C)
    xo, eo = xin.clone().requires_grad_(True), ein.clone().requires_grad_(True)
    want = R.grapher_forward(xo, pg, k=9, dilation=1, r=1, groups=G, training=True)
    want_e, _ = R.grapher_label_forward(eo, want, pl, k=9, groups=G, training=True)
    (want.sum() + want_e.square().sum()).backward()
    calls = {"fused": 0}
    real = fused._MRGroupedLinearBNAct.apply

    def counting(*a):
        calls["fused"] += 1
        return real(*a)
    fused._MRGroupedLinearBNAct.apply = counting
    try:
        _lib.prof_reset()
        _lib.prof_enable(True)
        xg, eg = xin.cuda().requires_grad_(True), ein.cuda().requires_grad_(True)
        out = g(xg)
        e2, _ = gl(eg, out)
        (out.sum() + e2.square().sum()).backward()
        torch.cuda.synchronize()
        _lib.prof_enable(False)
    finally:
        fused._MRGroupedLinearBNAct.apply = real
    assert calls["fused"] == 2
    assert _lib.prof_read()["mr_fwd"][1] == 2                    # the two fused launches are the only aggregation launches
    assert torch.allclose(out.cpu(), want, atol=1e-3, rtol=1e-3)
    assert torch.allclose(e2.cpu(), want_e, atol=1e-3, rtol=1e-3)
    assert torch.allclose(xg.grad.cpu(), xo.grad, atol=2e-3, rtol=2e-3)
    assert torch.allclose(eg.grad.cpu(), eo.grad, atol=2e-3, rtol=2e-3)
