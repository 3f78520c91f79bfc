"""GPU unit tests of the token-major bandwidth kernels (csrc/gkg_dense.hip) and the fused autograd Functions
against plain PyTorch fp32 references of the same ops (floating-point tolerance 1e-4 / 1e-3)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["vendor", "x6", "x6-deterministic"])
def gemm_mode(request):
    """Every projection test runs with the projections in the vendor GEMM library + separate BN passes ("vendor") and on the
    split-bf16 kernels — forward, input and weight gradient ("x6", the default; "-deterministic": the atomically accumulated x6
    weight gradient gives way to the library GEMM)."""
    from gkgnet_amd import fused
    old = (fused.DETERMINISTIC, fused.GEMM_MATH)
    fused.GEMM_MATH = request.param.split("-")[0]
    fused.DETERMINISTIC = request.param.endswith("deterministic")
    yield request.param
    fused.DETERMINISTIC, fused.GEMM_MATH = old


def _bn(C):
    torch.manual_seed(C)
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.1)
        bn.running_mean.normal_(0, 0.1); bn.running_var.uniform_(0.5, 1.5)
    return bn


@pytest.mark.parametrize("R,cin,cout,act,with_res", [(1000, 64, 32, 0, False), (777, 32, 64, 1, True),
                                                     (2560, 320, 1280, 1, False), (2560, 1280, 320, 0, True),
                                                     (10368, 320, 320, 0, False), (4100, 36, 40, 1, True),
                                                     (19, 16, 8, 1, False)])
def test_linear_bn_act_matches_torch(R, cin, cout, act, with_res, gemm_mode):
    from gkgnet_amd import fused
    torch.manual_seed(0)
    x = torch.randn(R, cin, device="cuda", requires_grad=True)
    conv = torch.nn.Conv2d(cin, cout, 1).cuda()
    bn = _bn(cout)
    res = torch.randn(R, cout, device="cuda", requires_grad=True) if with_res else None
    ref_bn = _bn(cout)
    ref_bn.load_state_dict(bn.state_dict())
    out = fused._LinearBNAct.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, res, bn, act, None)
    g = torch.randn_like(out)
    out.backward(g)
    assert conv.bias.grad is None            # identically zero, not materialised
    got = [out.detach(), x.grad, conv.weight.grad, torch.zeros_like(conv.bias), bn.weight.grad, bn.bias.grad]
    if with_res:
        got.append(res.grad)
    # reference: the reference's op chain Conv2d(1x1) -> BN(train) -> GELU on the (R,cin) rows viewed as (1,cin,R,1)
    x2 = x.detach().clone().requires_grad_(True)
    w2 = conv.weight.detach().clone().requires_grad_(True)
    b2 = conv.bias.detach().clone().requires_grad_(True)
    res2 = res.detach().clone().requires_grad_(True) if with_res else None
    y = F.conv2d(x2.t().reshape(1, cin, R, 1), w2, b2)
    y = F.batch_norm(y, ref_bn.running_mean, ref_bn.running_var, ref_bn.weight, ref_bn.bias, True, 0.1, 1e-5)
    if act:
        y = F.gelu(y)
    y = y.reshape(cout, R).t()
    if with_res:
        y = y + res2
    y.backward(g)
    want = [y.detach(), x2.grad, w2.grad, b2.grad, ref_bn.weight.grad, ref_bn.bias.grad]
    if with_res:
        want.append(res2.grad)
    names = ["out", "dx", "dW", "dbias", "dgamma", "dbeta", "dres"]
    for n, a, b in zip(names, got, want):
        tol = 2e-3 if n in ("dW", "dgamma", "dbeta", "dbias") else 2e-4
        assert torch.allclose(a, b, atol=tol, rtol=1e-3), (n, float((a - b).abs().max()))
    # running statistics follow nn.BatchNorm semantics (momentum 0.1, unbiased variance, conv bias included)
    assert torch.allclose(bn.running_mean, ref_bn.running_mean, atol=1e-5)
    assert torch.allclose(bn.running_var, ref_bn.running_var, atol=1e-4, rtol=1e-4)
    assert int(bn.num_batches_tracked) == 1


def test_grouped_linear_matches_grouped_conv(gemm_mode):
    """_GroupedLinearBNAct on the XM operand buffer == Conv2d(groups=4)+BN+GELU on the reference's interleaved (B,2C,N,1) input
    (torch_vertex.py:57-61 + torch_nn.py:57-69): output, input gradient (in the XM layout) and the weight gradient in the
    reference's column order."""
    from gkgnet_amd import fused
    from util import xm_interleaved
    torch.manual_seed(1)
    for R, C in ((900, 64), (1280, 80), (256, 48)):            # 2C channels, 4 groups of C / 2; chunks of 16 / 20 / 12
        XM = torch.randn(R, 2 * C, device="cuda", requires_grad=True)
        conv = torch.nn.Conv2d(2 * C, 2 * C, 1, groups=4).cuda()
        bn, ref_bn = _bn(2 * C), _bn(2 * C)
        ref_bn.load_state_dict(bn.state_dict())
        out = fused._GroupedLinearBNAct.apply(XM, conv.weight, conv.bias, bn.weight, bn.bias, bn, 1)
        g = torch.randn_like(out)
        out.backward(g)
        fused.flush_wgrads()
        XM2 = XM.detach().clone().requires_grad_(True)
        w2 = conv.weight.detach().clone().requires_grad_(True)
        xin = xm_interleaved(XM2).t().reshape(1, 2 * C, R, 1)
        y = F.gelu(F.batch_norm(F.conv2d(xin, w2, conv.bias.detach(), groups=4), ref_bn.running_mean, ref_bn.running_var,
                                ref_bn.weight, ref_bn.bias, True, 0.1, 1e-5))
        y = y.reshape(2 * C, R).t()
        y.backward(g)
        assert torch.allclose(out, y, atol=2e-4, rtol=1e-3)
        assert torch.allclose(XM.grad, XM2.grad, atol=2e-4, rtol=1e-3)
        assert torch.allclose(conv.weight.grad, w2.grad, atol=2e-3, rtol=1e-3)
        assert torch.allclose(bn.weight.grad, ref_bn.weight.grad, atol=2e-3, rtol=1e-3)
        assert torch.allclose(bn.bias.grad, ref_bn.bias.grad, atol=2e-3, rtol=1e-3)


def test_layout_round_trip_and_token_major_graph_ops():
    """to_token_major / its backward are exact transposes; the token-major k-NN and aggregation return exactly
    what the channel-major operators (already pinned to the oracle) return."""
    from gkgnet_amd import fused, ops
    torch.manual_seed(2)
    B, C, H, G, k = 3, 32, 10, 2, 5
    N = H * H
    x = torch.randn(B, C, H, H, device="cuda", requires_grad=True)
    xt = fused.to_token_major(x)
    assert torch.equal(xt, x.detach().permute(0, 2, 3, 1).reshape(B * N, C))
    xt.backward(xt.detach())
    assert torch.equal(x.grad, x.detach())
    rp = -torch.rand(1, N, N, device="cuda")
    xtm = xt.detach().view(B, N, C)
    edge_tm = fused.knn_graph_tm(xtm, None, rp, k, 2, G)
    edge_cm = ops.knn_graph(x.detach().reshape(B * G, C // G, N), None, rp, k, 2)
    assert torch.equal(edge_tm, edge_cm)
    m_cm = ops.max_relative(x.detach().reshape(B * G, C // G, N), edge_cm[0])            # (BG, c, N)
    m_tm = fused._MaxRelativeTM.apply(xtm, None, edge_tm[0], G, 0)                       # (B, N, C)
    assert torch.equal(m_tm.permute(0, 2, 1).reshape(B * G, C // G, N), m_cm)
    from util import xm_split
    XM = fused._MaxRelativeTM.apply(xtm, None, edge_tm[0], G, 1)                         # (T, 2C) operand buffer [x | m]
    xs, ms = xm_split(XM)
    assert torch.equal(xs, xtm.reshape(B * N, C)) and torch.equal(ms, m_tm.reshape(B * N, C))
    # x already in the buffer (the Grapher's fc1 writes it there): only the m chunks are written, same bits
    XM2 = torch.full((B * N, 2 * C), float("nan"), device="cuda")
    xv = fused._xm_xview(XM2, B, N, C)
    xv.copy_(xtm.view(B, N, 4, C // 4))
    e2 = fused.knn_graph_tm(xv, None, rp, k, 2, G)
    assert torch.equal(e2, edge_tm)
    e16 = fused.knn_graph_tm16(xv, None, rp, k, 2, G)
    assert torch.equal(e16.view(torch.uint16).to(torch.int64), edge_tm[0])
    XM3 = fused._MaxRelativeTM.apply(xv, None, edge_tm[0], G, 1)
    assert XM3.data_ptr() == XM2.data_ptr() and torch.equal(XM3, XM)
    XM2[:, :] = float("nan")
    xv.copy_(xtm.view(B, N, 4, C // 4))
    XM4 = fused._MaxRelativeTM.apply(xv, None, e16, G, 1)
    assert torch.equal(XM4, XM)


def test_bf16_outputs_are_the_rounded_fp32_outputs():
    """Inference under bf16 autocast writes GEMM-only intermediates as bf16 from the producing kernel: each must equal
    the fp32 result rounded to nearest-even, bit for bit (layout kernel, BN-apply(+GELU), aggregation operand)."""
    from gkgnet_amd import fused
    torch.manual_seed(3)
    B, C, H, G, k = 3, 64, 10, 4, 9
    N = H * H
    x = torch.randn(B, C, H, H, device="cuda")
    with torch.no_grad():
        t32, _ = fused._BlockEntry.apply(x, False)
        t16, _ = fused._BlockEntry.apply(x, True)
        assert t16.dtype == torch.bfloat16 and torch.equal(t16, t32.to(torch.bfloat16))
        assert torch.equal(t32.view(B, N, C), x.flatten(2).transpose(1, 2))
        # BN(eval)+GELU apply
        conv = torch.nn.Conv2d(C, 2 * C, 1).cuda()
        bn = _bn(2 * C).eval()
        o32 = fused._LinearBNAct.apply(t32, conv.weight, conv.bias, bn.weight, bn.bias, None, bn, 1, None, False)
        o16 = fused._LinearBNAct.apply(t32, conv.weight, conv.bias, bn.weight, bn.bias, None, bn, 1, None, True)
        assert o16.dtype == torch.bfloat16 and torch.equal(o16, o32.to(torch.bfloat16))
        ref = F.gelu(bn(conv(x))).flatten(2).transpose(1, 2).reshape(B * N, 2 * C)
        assert torch.allclose(o32, ref, atol=2e-5, rtol=1e-4)
        # aggregation operand XM (T, 2C) = [x | max-relative] chunks
        xb = t32.view(B, N, C)
        edge = fused.knn_graph_tm(xb, None, None, k, 1, G)
        for kk in (9, 5):                                     # specialised k = 9 and the generic-k kernel
            idx = edge[0][..., :kk].contiguous()
            u32 = fused._MaxRelativeTM.apply(xb, None, idx, G, 1, False)
            u16 = fused._MaxRelativeTM.apply(xb, None, idx, G, 1, True)
            assert u16.dtype == torch.bfloat16 and torch.equal(u16, u32.to(torch.bfloat16))
        # bf16 operands with fp32 result: same GEMM as an fp32 product of the rounded operands (fp32 accumulation)
        y = fused._mm_t(t16, conv.weight.view(2 * C, C))
        want = t16.float() @ conv.weight.view(2 * C, C).to(torch.bfloat16).float().t()
        assert y.dtype == torch.float32 and torch.allclose(y, want, atol=1e-3, rtol=1e-3)


def test_deterministic_weight_gradient_is_bit_reproducible():
    """fused.DETERMINISTIC: no atomically accumulated weight gradient (the split-K products are reduced by a library sum in
    a fixed order) -> two runs agree bit for bit, also with every projection forced onto the x6 kernels."""
    from gkgnet_amd import fused
    old = (fused.GEMM_MATH, fused.DETERMINISTIC)
    fused.GEMM_MATH, fused.DETERMINISTIC = "x6", True
    try:
        torch.manual_seed(4)
        x = torch.randn(5000, 96, device="cuda")
        conv = torch.nn.Conv2d(96, 160, 1).cuda()
        bn = _bn(160)
        g = torch.randn(5000, 160, device="cuda")
        grads = []
        for _ in range(2):
            conv.weight.grad = None
            out = fused._LinearBNAct.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, None, bn, 1, None)
            out.backward(g)
            grads.append(conv.weight.grad.clone())
        assert torch.equal(grads[0], grads[1])
    finally:
        fused.GEMM_MATH, fused.DETERMINISTIC = old


@pytest.mark.parametrize("mode", ["vendor", "x6"])
def test_bn_statistics_survive_a_large_channel_offset(mode):
    """Train-mode BN variance when |mean| >> std (y = 100 + 0.01*noise per channel): E[y^2] - E[y]^2 in fp32 would lose
    the variance entirely (1e-7 * 1e4 / 1e-4 = 10x its value); the statistics are taken centred (GEMM epilogue: per
    tile mean / M2 from registers, fp64 merge) or shifted by a sample row (stand-alone pass), so invstd stays within
    1e-3 of the fp64 value."""
    from gkgnet_amd import fused
    old = fused.GEMM_MATH
    fused.GEMM_MATH = mode
    try:
        torch.manual_seed(5)
        R, cin, cout = 6000, 32, 64
        x = torch.cat([torch.ones(R, 4, device="cuda"), 0.01 * torch.randn(R, cin - 4, device="cuda")], 1)
        conv = torch.nn.Conv2d(cin, cout, 1).cuda()
        with torch.no_grad():
            conv.weight[:, :4] = 25.0                      # y = 100 + O(0.01)
        bn = _bn(cout)
        with torch.no_grad():
            bn.weight.fill_(1.0); bn.bias.zero_()
        out = fused._LinearBNAct.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, None, bn, 0, None)
        y = (x.double() @ conv.weight.view(cout, cin).double().t())
        want = (y - y.mean(0)) / torch.sqrt(y.var(0, unbiased=False) + 1e-5)
        assert float(y.detach().mean().abs()) > 50 and float(y.detach().std(0).mean()) < 0.1
        # the fp32 product itself rounds y to ~6e-6 absolute, i.e. ~1e-3 of its 5e-3 standard deviation
        assert torch.allclose(out.detach().double(), want, atol=2e-2, rtol=1e-2), float((out.detach().double() - want).abs().max())
    finally:
        fused.GEMM_MATH = old


def test_deterministic_scatter_is_bit_reproducible_and_correct():
    """fused.DETERMINISTIC: the neighbour-gradient scatter of the max-relative backward adds each key's fan-in in a fixed
    order -> bit-identical from run to run, and equal (to rounding) to the default LDS-atomic kernel; self and bipartite
    graphs, the XM (mode 1) and plain (mode 0) gradient layouts, and the > 9 600-key global fallback."""
    from gkgnet_amd import fused
    torch.manual_seed(6)
    old = fused.DETERMINISTIC
    try:
        for (B, N, M, C, G, k, mode) in [(3, 200, None, 64, 4, 9, 1), (2, 80, 324, 64, 2, 9, 1), (2, 150, None, 32, 2, 5, 0),
                                         (1, 40, 12000, 16, 2, 9, 0)]:
            Mk = N if M is None else M
            x = torch.randn(B, N, C, device="cuda")
            src = None if M is None else torch.randn(B, Mk, C, device="cuda")
            # heavy fan-in: all queries choose among the first 7 keys
            idx = torch.randint(0, 7, (B * G, N, k), device="cuda")
            g = torch.randn((B * N, 2 * C) if mode == 1 else (B, N, C), device="cuda")
            outs = {}
            for det in (True, True, False):
                fused.DETERMINISTIC = det
                xg = x.clone().requires_grad_(True)
                sg = None if src is None else src.clone().requires_grad_(True)
                fused._MaxRelativeTM.apply(xg, sg, idx, G, mode).backward(g)
                outs.setdefault(det, []).append((xg.grad.clone(), None if sg is None else sg.grad.clone()))
            (a1, s1), (a2, s2) = outs[True]
            assert torch.equal(a1, a2) and (s1 is None or torch.equal(s1, s2))
            b1, t1 = outs[False][0]
            assert torch.allclose(a1, b1, atol=1e-4, rtol=1e-4)
            if s1 is not None:
                assert torch.allclose(s1, t1, atol=1e-4, rtol=1e-4)
    finally:
        fused.DETERMINISTIC = old


@pytest.mark.parametrize("act", [0, 1])
def test_affine_act_dual_writes_both_copies(act):
    """gkg_affine_act_dual: the fp32 result equals gkg_affine_act's and the second output is its bf16 rounding."""
    from gkgnet_amd import _lib
    lib = _lib.load()
    torch.manual_seed(act)
    R, C = 1003, 72
    Y, res = torch.randn(R, C, device="cuda"), torch.randn(R, C, device="cuda")
    a, c = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    scale = torch.rand(17, device="cuda")
    o1 = torch.empty(R, C, device="cuda"); o2 = torch.full((R, C), float("nan"), device="cuda")
    o16 = torch.full((R, C), float("nan"), device="cuda", dtype=torch.bfloat16)
    _lib.check(lib.gkg_affine_act(Y.data_ptr(), a.data_ptr(), c.data_ptr(), res.data_ptr(), o1.data_ptr(), R, C, 1, C, 0, 0, act,
                                  _lib.F32, scale.data_ptr(), 59, None), "gkg_affine_act")
    _lib.check(lib.gkg_affine_act_dual(Y.data_ptr(), a.data_ptr(), c.data_ptr(), res.data_ptr(), o2.data_ptr(), o16.data_ptr(),
                                       R, C, act, scale.data_ptr(), 59, None), "gkg_affine_act_dual")
    assert torch.equal(o1, o2) and torch.equal(o16, o1.bfloat16())


@pytest.mark.parametrize("train", [True, False])
def test_stem_and_downsample_bn_on_own_kernels_match_torch(train):
    """backbone.Stem / Downsample (reference gkgnet.py:74-118): 3x3 convolutions (library) + BN (+ GELU) — with the BN and the
    activation on the blocks' token-major kernels (fused.bn_act, channels-last) against the plain torch modules on the same
    weights: outputs, input gradient, every parameter gradient, running statistics."""
    from gkgnet_amd import fused, layers
    from gkgnet_amd.backbone import Downsample, Stem
    layers.norm_cfg["type"] = "BN"
    torch.manual_seed(11)
    for make, shape in ((lambda: Stem(out_dim=48, act="gelu"), (3, 3, 64, 64)), (lambda: Downsample(48, 96), (3, 48, 20, 20))):
        res = []
        for own in (True, False):
            torch.manual_seed(5)
            mod = make().cuda()
            mod.train(train)
            x = torch.randn(*shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)).requires_grad_(train)
            old = (fused.STEM_BN, fused.STEM_CONV)
            fused.STEM_BN = fused.STEM_CONV = own
            try:
                with torch.set_grad_enabled(train):
                    out = mod(x)
                    if train:
                        out.square().sum().backward()
            finally:
                fused.STEM_BN, fused.STEM_CONV = old
            res.append((out.detach(), x.grad, [p.grad for p in mod.parameters()], [b.clone() for b in mod.buffers()]))
        (o1, g1, p1, b1), (o2, g2, p2, b2) = res
        assert o1.shape == o2.shape and torch.allclose(o1, o2, atol=2e-4, rtol=2e-4), float((o1 - o2).abs().max())
        if train:
            assert torch.allclose(g1, g2, atol=2e-3, rtol=2e-3), float((g1 - g2).abs().max())
            for a, b in zip(p1, p2):
                assert torch.allclose(a, b, atol=2e-3 * max(1.0, float(b.abs().max())), rtol=2e-3), float((a - b).abs().max())
        for a, b in zip(b1, b2):
            assert torch.allclose(a.float(), b.float(), atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("cout", [24, 40, 48, 64])
@pytest.mark.parametrize("shape", [(2, 3, 64, 64), (1, 3, 37, 51), (3, 4, 20, 33)])
def test_stem_first_convolution_direct_kernel(cout, shape):
    """csrc/gkg_stem.hip against F.conv2d (reference gkgnet.py:79-81: Conv2d(3 -> C1/2, 3, stride 2, padding 1)): the plain
    convolution with its autograd (weight / bias gradients from the library's convolution backward), and the inference form
    with eval-mode BN + GELU folded, fp32 and bf16 outputs; odd image sizes (the last row / column of taps is padding)."""
    import torch.nn.functional as F
    from gkgnet_amd import fused, layers
    layers.norm_cfg["type"] = "BN"
    B, cin, H, W = shape
    torch.manual_seed(cout + H)
    conv = torch.nn.Conv2d(cin, cout, 3, stride=2, padding=1).cuda()
    bn = layers.build_norm(cout).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    x = torch.randn(B, cin, H, W, device="cuda")
    assert fused.stem_conv_supported(conv, x)
    y = fused.stem_conv(conv, x)
    want = F.conv2d(x, conv.weight, conv.bias, stride=2, padding=1)
    assert y.shape == want.shape and y.permute(0, 2, 3, 1).is_contiguous()
    assert torch.allclose(y, want, atol=2e-5, rtol=2e-5), float((y - want).abs().max())
    # gradients
    cot = torch.randn_like(want)
    y.backward(cot)
    gw, gb = conv.weight.grad.clone(), conv.bias.grad.clone()
    conv.zero_grad()
    F.conv2d(x, conv.weight, conv.bias, stride=2, padding=1).backward(cot)
    assert torch.allclose(gw, conv.weight.grad, atol=1e-3, rtol=1e-3) and torch.allclose(gb, conv.bias.grad, atol=1e-3, rtol=1e-3)
    # inference: conv + BN(eval) + GELU in one launch
    ref = F.gelu(bn(want))
    got32 = fused.stem_conv_bn_act_eval(conv, bn, torch.nn.GELU(), x, False)
    assert torch.allclose(got32, ref, atol=5e-5, rtol=5e-5), float((got32 - ref).abs().max())
    got16 = fused.stem_conv_bn_act_eval(conv, bn, torch.nn.GELU(), x, True)
    assert got16.dtype == torch.bfloat16 and torch.allclose(got16.float(), ref, atol=2e-2, rtol=1e-2)
    lin = fused.stem_conv_bn_act_eval(conv, bn, None, x, False)
    assert torch.allclose(lin, bn(want), atol=5e-5, rtol=5e-5)
