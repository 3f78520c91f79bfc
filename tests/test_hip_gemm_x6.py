"""The split-bf16 ("x6") projection kernels through the C ABI (csrc/gkg_gemm_x6.hip): forward with BN statistics and the
input gradient, against fp64 evaluations.  The accuracy bar is the fp32 one: the error must not exceed what a plain fp32
GEMM (torch) makes on the same operands — measured it is 3-4x smaller."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _planes(lib, w, nb, cout, cin, kperm=0):
    """Forward / dgrad planes of w (nb, cout, cin) through the batched prep entry points."""
    pf = torch.empty(lib.gkg_x6_planes_bytes(cin, cout, nb, 0), dtype=torch.uint8, device="cuda")
    pd = torch.empty(lib.gkg_x6_planes_bytes(cin, cout, nb, 1), dtype=torch.uint8, device="cuda")
    host = ctypes.create_string_buffer(lib.gkg_x6_prep_desc_bytes())
    units = lib.gkg_x6_prep_desc_fill(host, 0, w.data_ptr(), pf.data_ptr(), pd.data_ptr(), cin, cout, nb, 0, kperm)
    assert units > 0
    descs = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).cuda()
    assert lib.gkg_x6_prep_weights(descs.data_ptr(), 1, units, None) == 0
    torch.cuda.synchronize()
    return pf, pd


def _rel(a, ref, scale):
    return float(((a.double() - ref).abs() / scale).max())


# (R, cin, cout, nb): cfg2's layers, ragged rows, K not a multiple of 32 (36, 400), N below one tile, grouped (nb = 4)
SHAPES = [(10368, 320, 320, 1), (2560, 320, 1280, 1), (2560, 1280, 320, 1), (10368, 160, 160, 4), (777, 36, 40, 1),
          (4100, 400, 400, 1), (129, 64, 8, 2), (19, 16, 8, 1), (128, 32, 64, 1),
          # 80 output columns on >= 10 240 rows: the 80-column tile form (forward of cout = 80, dgrad of cin = 80), ragged rows,
          # K = 100 (tail step), grouped
          (10368, 80, 80, 1), (12001, 160, 80, 1), (10300, 80, 320, 1), (11000, 100, 80, 1), (10250, 80, 80, 2)]


@pytest.mark.parametrize("R,cin,cout,nb", SHAPES)
def test_forward_and_dgrad_at_fp32_accuracy(R, cin, cout, nb):
    from gkgnet_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(R * 7 + cin)
    x = torch.randn(nb, R, cin, device="cuda", generator=gen) * 2
    w = torch.randn(nb, cout, cin, device="cuda", generator=gen) * 0.1
    dy = torch.randn(nb, R, cout, device="cuda", generator=gen)
    pf, pd = _planes(lib, w, nb, cout, cin)
    y = torch.full((nb, R, cout), float("nan"), device="cuda")
    _lib.check(lib.gkg_linear_bn_fwd_x6(x.data_ptr(), cin, R * cin, pf.data_ptr(), y.data_ptr(), R, cin, cout, nb, 0,
                                        *([None] * 10), 0.0, 0.0, None, None), "fwd")
    dx = torch.full((nb, R, cin), float("nan"), device="cuda")
    _lib.check(lib.gkg_linear_dgrad_x6(dy.data_ptr(), cout, R * cout, pd.data_ptr(), dx.data_ptr(), R, cin, cout, nb, None),
               "dgrad")
    ref_y = torch.bmm(x.double(), w.double().transpose(1, 2))
    ref_dx = torch.bmm(dy.double(), w.double())
    mag_y = torch.bmm(x.double().abs(), w.double().abs().transpose(1, 2)) + 1e-30      # sum |a b|: the fp32 bound's scale
    mag_dx = torch.bmm(dy.double().abs(), w.double().abs()) + 1e-30
    e_y, e_dx = _rel(y, ref_y, mag_y), _rel(dx, ref_dx, mag_dx)
    f_y = _rel(torch.bmm(x, w.transpose(1, 2)), ref_y, mag_y)                         # what plain fp32 makes of it
    f_dx = _rel(torch.bmm(dy, w), ref_dx, mag_dx)
    assert e_y <= max(f_y, 1.2e-7) and e_dx <= max(f_dx, 1.2e-7), (e_y, f_y, e_dx, f_dx)
    assert e_y < 2e-7 and e_dx < 2e-7                                                   # 2^-23 .. 2^-22 of sum |a b|


WSHAPES = [(10368, 320, 320, 1), (10368, 160, 160, 4), (2560, 1280, 320, 1), (20000, 80, 80, 1), (777, 36, 40, 1),
           (4100, 400, 100, 1), (129, 64, 8, 2), (19, 16, 8, 1), (3000, 1, 3, 1)]


@pytest.mark.parametrize("R,cin,cout,nb", WSHAPES)
def test_wgrad_at_fp32_accuracy(R, cin, cout, nb):
    from gkgnet_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(R * 3 + cout)
    x = torch.randn(nb, R, cin, device="cuda", generator=gen) * 2
    dy = torch.randn(nb, R, cout, device="cuda", generator=gen)
    dw = torch.zeros(nb, cout, cin, device="cuda")
    _lib.check(lib.gkg_linear_wgrad_x6(dy.data_ptr(), cout, R * cout, x.data_ptr(), cin, R * cin, dw.data_ptr(), R, cin, cout,
                                       nb, 0, None), "wgrad")
    ref = torch.bmm(dy.double().transpose(1, 2), x.double())
    mag = torch.bmm(dy.double().abs().transpose(1, 2), x.double().abs()) + 1e-30
    e = _rel(dw, ref, mag)
    f = _rel(torch.bmm(dy.transpose(1, 2), x), ref, mag)
    assert e <= max(f, 1.2e-7) and e < 2e-7, (e, f)


def test_wgrad_column_slices():
    """dy / x as column slices of wider token-major matrices (the grouped projection's operands)."""
    from gkgnet_amd import _lib
    lib = _lib.load()
    torch.manual_seed(4)
    R, cin, cout, nb = 1500, 24, 40, 4
    xw = torch.randn(R, nb * cin, device="cuda")
    dyw = torch.randn(R, nb * cout, device="cuda")
    dw = torch.zeros(nb, cout, cin, device="cuda")
    _lib.check(lib.gkg_linear_wgrad_x6(dyw.data_ptr(), nb * cout, cout, xw.data_ptr(), nb * cin, cin, dw.data_ptr(), R, cin,
                                       cout, nb, 0, None), "wgrad")
    want = torch.einsum("rqn,rqk->qnk", dyw.view(R, nb, cout).double(), xw.view(R, nb, cin).double())
    assert torch.allclose(dw.double(), want, atol=1e-4, rtol=1e-5)


def test_column_slices_and_row_pitch():
    """x as a column slice of a wider token-major matrix (the grouped projection's operand) and dy with a row pitch."""
    from gkgnet_amd import _lib
    lib = _lib.load()
    torch.manual_seed(3)
    R, cin, cout, nb = 900, 32, 48, 4
    wide = torch.randn(R, nb * cin, device="cuda")                 # group q = columns [q*cin, (q+1)*cin)
    w = torch.randn(nb, cout, cin, device="cuda") * 0.2
    pf, pd = _planes(lib, w, nb, cout, cin)
    y = torch.empty(nb, R, cout, device="cuda")
    _lib.check(lib.gkg_linear_bn_fwd_x6(wide.data_ptr(), nb * cin, cin, pf.data_ptr(), y.data_ptr(), R, cin, cout, nb, 0,
                                        *([None] * 10), 0.0, 0.0, None, None), "fwd")
    want = torch.einsum("rqk,qnk->qrn", wide.view(R, nb, cin).double(), w.double())
    assert torch.allclose(y.double(), want, atol=1e-5, rtol=1e-5)
    dyw = torch.randn(R, nb * cout, device="cuda")
    dx = torch.empty(nb, R, cin, device="cuda")
    _lib.check(lib.gkg_linear_dgrad_x6(dyw.data_ptr(), nb * cout, cout, pd.data_ptr(), dx.data_ptr(), R, cin, cout, nb, None),
               "dgrad")
    want = torch.einsum("rqn,qnk->qrk", dyw.view(R, nb, cout).double(), w.double())
    assert torch.allclose(dx.double(), want, atol=1e-5, rtol=1e-5)


def test_train_statistics_epilogue():
    """train = 1: BN scale / shift / saved statistics / running statistics from the epilogue's fp64 column sums."""
    _check_train_statistics(3001, 96, 72)
    _check_train_statistics(12001, 96, 80)                           # the 80-column tile form


def _check_train_statistics(R, cin, cout):
    from gkgnet_amd import _lib, fused
    lib = _lib.load()
    torch.manual_seed(5)
    x = torch.randn(R, cin, device="cuda") + 3.0                     # a large mean against the spread
    w = torch.randn(cout, cin, device="cuda") * 0.1
    pf, _ = _planes(lib, w, 1, cout, cin)
    gamma, beta, bias = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda"), torch.randn(cout, device="cuda")
    rm, rv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
    nbt = torch.zeros((), dtype=torch.int64, device="cuda")
    a, c, mean, invstd = (torch.empty(cout, device="cuda") for _ in range(4))
    y = torch.empty(R, cout, device="cuda")
    stats = fused._stats_scratch(x.device)
    _lib.check(lib.gkg_linear_bn_fwd_x6(x.data_ptr(), cin, R * cin, pf.data_ptr(), y.data_ptr(), R, cin, cout, 1, 1,
                                        gamma.data_ptr(), beta.data_ptr(), bias.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                                        nbt.data_ptr(), a.data_ptr(), c.data_ptr(), mean.data_ptr(), invstd.data_ptr(), 0.1,
                                        1e-5, stats.data_ptr(), None), "fwd train")
    yd = x.double() @ w.double().t()
    m, v = yd.mean(0), yd.var(0, unbiased=False)
    assert torch.allclose(mean.double(), m, atol=1e-5) and torch.allclose(invstd.double(), (v + 1e-5).rsqrt(), rtol=1e-5)
    assert torch.allclose(a.double(), gamma.double() * (v + 1e-5).rsqrt(), rtol=1e-5)
    assert torch.allclose(c.double(), beta.double() - gamma.double() * (v + 1e-5).rsqrt() * m, atol=2e-5, rtol=1e-5)
    assert torch.allclose(rm.double(), 0.1 * (m + bias.double()), atol=1e-5)
    assert torch.allclose(rv.double(), 0.9 + 0.1 * yd.var(0, unbiased=True), rtol=1e-5)
    assert int(nbt) == 1 and float(stats.abs().max()) == 0.0          # the scratch is clean again


def test_planes_follow_the_parameter(monkeypatch):
    """The fused layer re-splits its weight after an in-place update (optimiser step) and after the storage is replaced."""
    from gkgnet_amd import fused
    monkeypatch.setattr(fused, "GEMM_MATH", "x6")
    torch.manual_seed(9)
    R, cin, cout = 640, 64, 64
    conv = torch.nn.Conv2d(cin, cout, 1).cuda()
    bn = torch.nn.BatchNorm2d(cout).cuda().train()
    x = torch.randn(R, cin, device="cuda")

    def run():
        return fused._LinearBNAct.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, None, bn, 0, None)

    def want():
        y = x.double() @ conv.weight.detach().double().view(cout, cin).t()
        return ((y - y.mean(0)) * (y.var(0, unbiased=False) + bn.eps).rsqrt() * bn.weight.detach().double()
                + bn.bias.detach().double())

    assert torch.allclose(run().double(), want(), atol=1e-4)
    with torch.no_grad():
        conv.weight.mul_(-0.5).add_(0.01)                              # in place: version counter moves
    assert torch.allclose(run().double(), want(), atol=1e-4)
    with torch.no_grad():
        conv.weight.data = torch.randn_like(conv.weight) * 0.1         # new storage
    assert torch.allclose(run().double(), want(), atol=1e-4)


def test_captured_step_resplits_the_weights(monkeypatch):
    """A hipGraph-captured step sees weight updates made between replays (the capture holds the refresh launch)."""
    from gkgnet_amd import fused
    monkeypatch.setattr(fused, "GEMM_MATH", "x6")
    torch.manual_seed(11)
    R, cin, cout = 512, 32, 32
    conv = torch.nn.Conv2d(cin, cout, 1).cuda()
    bn = torch.nn.BatchNorm2d(cout).cuda().train()
    x = torch.randn(R, cin, device="cuda")

    def step():
        with torch.no_grad():
            return fused._LinearBNAct.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, None, bn, 0, None)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()                                                     # eager warm-up registers the weight
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    with torch.no_grad():
        conv.weight.copy_(torch.randn_like(conv.weight) * 0.3)
    g.replay()
    torch.cuda.synchronize()
    y = x.double() @ conv.weight.detach().double().view(cout, cin).t()
    want = (y - y.mean(0)) * (y.var(0, unbiased=False) + bn.eps).rsqrt() * bn.weight.detach().double() + bn.bias.detach().double()
    assert torch.allclose(out.double(), want, atol=1e-4)


def test_eager_calls_between_replays_of_a_step_that_updates_the_weights(monkeypatch):
    """capture (forward + in-graph weight update), replay, eager, replay, eager — every eager call must see the
    weights as the last replay left them (no version counter moves inside a replay, so the planes cannot be trusted once a
    capture exists)."""
    from gkgnet_amd import fused
    monkeypatch.setattr(fused, "GEMM_MATH", "x6")
    torch.manual_seed(12)
    R, cin, cout = 384, 32, 48
    conv = torch.nn.Conv2d(cin, cout, 1).cuda()
    bn = torch.nn.BatchNorm2d(cout).cuda().train()
    x = torch.randn(R, cin, device="cuda")
    delta = torch.randn_like(conv.weight) * 0.25

    def fwd():
        with torch.no_grad():
            return fused._LinearBNAct.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, None, bn, 0, None)

    def want():
        y = x.double() @ conv.weight.detach().double().view(cout, cin).t()
        return (y - y.mean(0)) * (y.var(0, unbiased=False) + bn.eps).rsqrt() * bn.weight.detach().double() + bn.bias.detach().double()

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fwd()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fwd()
        conv.weight.data.add_(delta)                    # the "optimiser step" inside the graph: AFTER the in-graph re-split
    for it in range(3):
        w_before = conv.weight.detach().clone()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(conv.weight.detach(), w_before + delta)
        got = fwd()                                     # eager, on the weights the replay left behind
        torch.cuda.synchronize()
        assert torch.allclose(got.double(), want(), atol=1e-4), it


def test_grouped_weight_gradient_on_the_streaming_kernel(monkeypatch):
    """fused._wgrad_grouped through gkg_linear_wgrad_x6 with a batch of groups (the grouped 1x1 projection's dW), written
    into a caller-provided slot, against an fp64 evaluation; plain and with the XM operand view + column permutation."""
    from gkgnet_amd import fused
    torch.manual_seed(21)
    nb, R, co, ci = 4, 2304 + 40, 160, 96                     # whole 128-row units + a ragged rest
    dY = torch.randn(nb, R, co, device="cuda")
    U = torch.randn(nb, R, ci, device="cuda")
    want = torch.bmm(dY.double().transpose(1, 2), U.double())
    monkeypatch.setattr(fused, "GEMM_MATH", "x6")
    slot = torch.full((nb, co, ci), 7.0, device="cuda")      # stale contents must not leak into the sum
    got = fused._wgrad_grouped(dY, U, slot)
    assert got.data_ptr() == slot.data_ptr()
    scale = (dY.double().abs().transpose(1, 2) @ U.double().abs()).max()
    assert float((got.double() - want).abs().max() / scale) < 2e-6
    assert fused._x6_wgrad_ok(dY, U, nb) is True and fused._x6_wgrad_ok(dY[0], U[0]) is True
    # the same problem with U as the XM operand buffer's view and kperm: dW comes back in the reference's interleaved columns
    XM = U.permute(1, 0, 2).reshape(R, nb * ci).contiguous()
    Uv = XM.view(R, nb, ci).permute(1, 0, 2)
    slot2 = torch.full((nb, co, ci), 7.0, device="cuda")
    got2 = fused._wgrad_grouped(dY, Uv, slot2, kperm=1)
    want2 = want.view(nb, co, 2, ci // 2).permute(0, 1, 3, 2).reshape(nb, co, ci)
    assert float((got2.double() - want2).abs().max() / scale) < 2e-6
    monkeypatch.setattr(fused, "DETERMINISTIC", True)
    got3 = fused._wgrad_grouped(dY, Uv, None, kperm=1)       # the library path: un-permuted by _kperm_grad_back
    assert float((got3.double() - want2).abs().max() / scale) < 2e-6
    assert fused._x6_wgrad_ok(dY[0], U[0]) is False           # fp32 atomics: never under GKG_DETERMINISTIC


@pytest.mark.parametrize("R,cin,cout,nb", [(1000, 256, 100, 1),      # 64 x 128 tiles, cin on the wide side, ragged rows
                                            (1000, 320, 256, 1),      # cout on the wide side: operands exchanged, dW transposed
                                            (2600, 640, 320, 1), (2600, 320, 1280, 1), (777, 128, 128, 3),
                                            (1000, 320, 320, 1)])     # 20 % padding: stays on 64 x 64 tiles
def test_weight_gradient_tile_shapes(R, cin, cout, nb):
    """gkg_linear_wgrad_x6 over the shapes that select each of its tile forms, against fp64."""
    from gkgnet_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(R + cin)
    dy = torch.randn(nb, R, cout, device="cuda", generator=gen)
    x = torch.randn(nb, R, cin, device="cuda", generator=gen)
    dw = torch.zeros(nb, cout, cin, device="cuda")
    _lib.check(lib.gkg_linear_wgrad_x6(dy.data_ptr(), cout, R * cout, x.data_ptr(), cin, R * cin, dw.data_ptr(), R, cin, cout,
                                       nb, 0, None), "gkg_linear_wgrad_x6")
    torch.cuda.synchronize()
    want = torch.bmm(dy.double().transpose(1, 2), x.double())
    scale = torch.bmm(dy.double().abs().transpose(1, 2), x.double().abs()).max()
    assert float((dw.double() - want).abs().max() / scale) < 2e-6


def test_bn_scratch_protocol_under_capture_replay_and_eager_interleaving():
    """The two-launch BN passes accumulate fp64 sums in alternating scratch buffers whose host-side cursor cannot see
    replays: a captured forward + backward with an ODD number of BN passes (3 layers -> 3 forward + 3 backward... plus one
    extra forward = 7), replayed, followed by eager steps, replayed again — every result must equal the eager-only one."""
    from gkgnet_amd import fused
    torch.manual_seed(13)
    R, C = 640, 64
    convs = [torch.nn.Conv2d(C, C, 1).cuda() for _ in range(3)]
    bns = [torch.nn.BatchNorm2d(C).cuda().train() for _ in range(3)]
    x = torch.randn(R, C, device="cuda", requires_grad=True)
    params = [p for m in convs + bns for p in m.parameters()]

    def step():
        h = x
        for conv, bn in zip(convs, bns):
            h = fused._LinearBNAct.apply(h, conv.weight, conv.bias, bn.weight, bn.bias, None, bn, 1, None)
        with torch.no_grad():                            # a seventh BN pass: odd count per step
            fused._LinearBNAct.apply(h.detach(), convs[0].weight, convs[0].bias, bns[0].weight, bns[0].bias, None, bns[0], 0, None)
        grads = torch.autograd.grad(h.square().mean(), [x] + params, allow_unused=True)   # conv biases: folded, no gradient
        return [h.detach()] + [g for g in grads if g is not None]

    def snapshot(vals):
        return [v.detach().clone() for v in vals]

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            ref = snapshot(step())                       # eager reference (batch statistics: identical every time)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        outs = step()

    def check(vals, what):
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(vals, ref)):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), (what, i, float((a - b).abs().max()))

    for it in range(3):
        g.replay()
        check(outs, f"replay {it}")
        g.replay()                                      # two replays back to back
        check(outs, f"second replay {it}")
        for j in range(it + 1):                         # 1, 2, 3 eager steps in between: both cursor parities
            check(step(), f"eager {it}.{j}")


def test_random_shapes_forward_dgrad_wgrad():
    """Seeded sweep over ragged shapes (rows / channels not multiples of the tile sizes, K tails, batches)."""
    from gkgnet_amd import _lib
    lib = _lib.load()
    rng = np.random.RandomState(20260)
    for it in range(40):
        R = int(rng.choice([1, 7, 31, 128, 129, 500, 1000, 4097]))
        cin = 4 * int(rng.randint(1, 90))
        cout = 4 * int(rng.randint(1, 90))
        nb = int(rng.choice([1, 1, 2, 4]))
        gen = torch.Generator(device="cuda").manual_seed(it)
        x = torch.randn(nb, R, cin, device="cuda", generator=gen)
        w = torch.randn(nb, cout, cin, device="cuda", generator=gen) * 0.2
        dy = torch.randn(nb, R, cout, device="cuda", generator=gen)
        pf, pd = _planes(lib, w, nb, cout, cin)
        y = torch.full((nb, R, cout), float("nan"), device="cuda")
        dx = torch.full((nb, R, cin), float("nan"), device="cuda")
        dw = torch.zeros(nb, cout, cin, device="cuda")
        _lib.check(lib.gkg_linear_bn_fwd_x6(x.data_ptr(), cin, R * cin, pf.data_ptr(), y.data_ptr(), R, cin, cout, nb, 0,
                                            *([None] * 10), 0.0, 0.0, None, None), "fwd")
        _lib.check(lib.gkg_linear_dgrad_x6(dy.data_ptr(), cout, R * cout, pd.data_ptr(), dx.data_ptr(), R, cin, cout, nb, None),
                   "dgrad")
        _lib.check(lib.gkg_linear_wgrad_x6(dy.data_ptr(), cout, R * cout, x.data_ptr(), cin, R * cin, dw.data_ptr(), R, cin,
                                           cout, nb, 0, None), "wgrad")
        tag = (it, R, cin, cout, nb)
        assert torch.allclose(y.double(), torch.bmm(x.double(), w.double().transpose(1, 2)), atol=2e-5, rtol=1e-5), tag
        assert torch.allclose(dx.double(), torch.bmm(dy.double(), w.double()), atol=2e-5, rtol=1e-5), tag
        assert torch.allclose(dw.double(), torch.bmm(dy.double().transpose(1, 2), x.double()), atol=2e-4, rtol=1e-5), tag


def test_non_finite_inputs_poison_only_their_rows():
    """A NaN / inf in one activation row reaches that output row only (the split turns inf into NaN: hi = inf, residual
    inf - inf); the K tail of the row before it and the rows of other tiles stay finite."""
    from gkgnet_amd import _lib
    lib = _lib.load()
    torch.manual_seed(2)
    R, cin, cout = 300, 36, 40                     # K = 36: the last K-step is a tail
    x = torch.randn(1, R, cin, device="cuda")
    x[0, 17, 3] = float("nan")
    x[0, 200, 35] = float("inf")
    x[0, 201, 0] = float("nan")                    # the element right after row 200's tail in memory
    w = torch.randn(1, cout, cin, device="cuda")
    pf, _ = _planes(lib, w, 1, cout, cin)
    y = torch.empty(1, R, cout, device="cuda")
    _lib.check(lib.gkg_linear_bn_fwd_x6(x.data_ptr(), cin, R * cin, pf.data_ptr(), y.data_ptr(), R, cin, cout, 1, 0,
                                        *([None] * 10), 0.0, 0.0, None, None), "fwd")
    bad = ~torch.isfinite(y[0]).all(dim=1)
    assert bad.nonzero().flatten().tolist() == [17, 200, 201]


# ---- split-K forms (round 5): few rows under a long contraction.  (R, cin, cout, nb): the label branch's products, ragged rows,
# a contraction that is not a multiple of 32 (its last range ends in a tail step), grouped, and one shape the rule does not split
SK_SHAPES = [(2560, 1280, 320, 1), (2560, 640, 320, 1), (2560, 320, 320, 1), (2560, 320, 640, 1), (300, 2048, 64, 2),
             (1000, 1000, 100, 1), (777, 516, 40, 3), (2560, 320, 1280, 1)]


@pytest.mark.parametrize("R,cin,cout,nb", SK_SHAPES)
def test_split_k_forward_statistics_and_dgrad(R, cin, cout, nb):
    """gkg_linear_bn_fwd_x6_sk / gkg_linear_dgrad_x6_sk against fp64 at the fp32 bar, run-to-run identical bits (the partials
    are added in range order by the last arrival), BN column sums from the summed tile, counters left zero."""
    from gkgnet_amd import _lib, fused
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(R + cin + cout)
    x = torch.randn(nb, R, cin, device="cuda", generator=gen) * 2 + 0.5
    w = torch.randn(nb, cout, cin, device="cuda", generator=gen) * 0.1
    dy = torch.randn(nb, R, cout, device="cuda", generator=gen)
    pf, pd = _planes(lib, w, nb, cout, cin)
    ws = torch.zeros(lib.gkg_x6_splitk_workspace_bytes(), dtype=torch.uint8, device="cuda")
    stats = torch.zeros(lib.gkg_linear_stats_doubles(), dtype=torch.float64, device="cuda")
    ys, dxs = [], []
    for rep in range(2):
        y = torch.full((nb, R, cout), float("nan"), device="cuda")
        stats.zero_()
        _lib.check(lib.gkg_linear_bn_fwd_x6_sk(x.data_ptr(), cin, R * cin, pf.data_ptr(), y.data_ptr(), R, cin, cout, nb, 2,
                                               *([None] * 10), 0.0, 0.0, stats.data_ptr(), ws.data_ptr(), ws.numel(), 0, None), "fwd sk")
        dx = torch.full((nb, R, cin), float("nan"), device="cuda")
        res = torch.randn(nb, R, cin, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9))
        _lib.check(lib.gkg_linear_dgrad_x6_sk(dy.data_ptr(), cout, R * cout, pd.data_ptr(), dx.data_ptr(), R, cin, cout, nb,
                                              res.data_ptr(), ws.data_ptr(), ws.numel(), 0, 0, 0, None), "dgrad sk")
        torch.cuda.synchronize()
        ys.append(y)
        dxs.append(dx)
        assert int(ws[:4096].view(torch.int32).abs().max()) == 0                     # every tile counter re-armed
    assert torch.equal(ys[0], ys[1]) and torch.equal(dxs[0], dxs[1])
    y, dx = ys[0], dxs[0]
    ref_y = torch.bmm(x.double(), w.double().transpose(1, 2))
    ref_dx = torch.bmm(dy.double(), w.double()) + res.double()                      # the residual rides in the epilogue
    mag_y = torch.bmm(x.double().abs(), w.double().abs().transpose(1, 2)) + 1e-30
    mag_dx = torch.bmm(dy.double().abs(), w.double().abs()) + res.double().abs() + 1e-30
    e_y, e_dx = _rel(y, ref_y, mag_y), _rel(dx, ref_dx, mag_dx)
    f_y = _rel(torch.bmm(x, w.transpose(1, 2)), ref_y, mag_y)
    f_dx = _rel(torch.baddbmm(res, dy, w), ref_dx, mag_dx)
    assert e_y <= max(f_y, 1.2e-7) and e_dx <= max(f_dx, 1.2e-7), (e_y, f_y, e_dx, f_dx)
    assert e_y < 2e-7 and e_dx < 2e-7
    # the epilogue's column sums [nb][2][cout]: sum y, sum y^2
    sums = stats[:nb * 2 * cout].view(nb, 2, cout)
    assert torch.allclose(sums[:, 0], ref_y.sum(1), rtol=1e-6, atol=1e-3)
    assert torch.allclose(sums[:, 1], (ref_y * ref_y).sum(1), rtol=1e-5)
    # the same bits as the unsplit kernel would be a coincidence (different summation tree); the same VALUES to fp32 accuracy
    y1 = torch.empty_like(y)
    _lib.check(lib.gkg_linear_bn_fwd_x6(x.data_ptr(), cin, R * cin, pf.data_ptr(), y1.data_ptr(), R, cin, cout, nb, 0,
                                        *([None] * 10), 0.0, 0.0, None, None), "fwd")
    assert float(((y1.double() - y.double()).abs() / mag_y).max()) < 3e-7


# ---- few rows: the waves of a workgroup split K (gemm_x6_ks_kernel, round 5).  The _sk entry points take it for <= 4 096 rows.
KS_SHAPES = [(2560, 320, 320, 1), (2560, 1280, 320, 1), (2560, 320, 1280, 1), (2560, 160, 160, 4), (2560, 640, 320, 1),
             (777, 36, 40, 1), (129, 64, 8, 2), (19, 16, 8, 1), (4096, 100, 72, 1), (33, 2048, 64, 3), (300, 72, 200, 2)]


@pytest.mark.parametrize("R,cin,cout,nb", KS_SHAPES)
def test_k_split_inside_the_workgroup_forward_statistics_dgrad(R, cin, cout, nb):
    """Forward (+ BN column sums), input gradient (+ residual) on the K-split-in-workgroup body against fp64 at the fp32 bar and
    against gemm_x6_kernel (GKG_X6_NO_KS) on the same operands; run-to-run identical bits (partials are added in wave order)."""
    from gkgnet_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(R * 5 + cin + cout)
    x = torch.randn(nb, R, cin, device="cuda", generator=gen) * 2 + 0.5
    w = torch.randn(nb, cout, cin, device="cuda", generator=gen) * 0.1
    dy = torch.randn(nb, R, cout, device="cuda", generator=gen)
    res = torch.randn(nb, R, cin, device="cuda", generator=gen)
    pf, pd = _planes(lib, w, nb, cout, cin)
    ws = torch.zeros(lib.gkg_x6_splitk_workspace_bytes(), dtype=torch.uint8, device="cuda")
    stats = torch.zeros(lib.gkg_linear_stats_doubles(), dtype=torch.float64, device="cuda")
    out = {}
    # the K-split body forced for every short matrix, twice (GKG_X6_FORCE_KS, a per-call flag); then gemm_x6_kernel (GKG_X6_NO_KS)
    for flags in (_lib.X6_FORCE_KS, _lib.X6_FORCE_KS, _lib.X6_NO_KS):
        y = torch.full((nb, R, cout), float("nan"), device="cuda")
        stats.zero_()
        _lib.check(lib.gkg_linear_bn_fwd_x6_sk(x.data_ptr(), cin, R * cin, pf.data_ptr(), y.data_ptr(), R, cin, cout, nb, 2,
                                               *([None] * 10), 0.0, 0.0, stats.data_ptr(), ws.data_ptr(), ws.numel(), flags, None), "fwd")
        dx = torch.full((nb, R, cin), float("nan"), device="cuda")
        _lib.check(lib.gkg_linear_dgrad_x6_sk(dy.data_ptr(), cout, R * cout, pd.data_ptr(), dx.data_ptr(), R, cin, cout, nb,
                                              res.data_ptr(), ws.data_ptr(), ws.numel(), 0, 0, flags, None), "dgrad")
        torch.cuda.synchronize()
        out.setdefault(flags, []).append((y, dx, stats[:nb * 2 * cout].clone()))
    (y, dx, st), (y_b, dx_b, _), (y_ref, dx_ref, st_ref) = out[_lib.X6_FORCE_KS][0], out[_lib.X6_FORCE_KS][1], out[_lib.X6_NO_KS][0]
    assert torch.equal(y, y_b) and torch.equal(dx, dx_b)
    ref_y = torch.bmm(x.double(), w.double().transpose(1, 2))
    ref_dx = torch.bmm(dy.double(), w.double()) + res.double()
    mag_y = torch.bmm(x.double().abs(), w.double().abs().transpose(1, 2)) + 1e-30
    mag_dx = torch.bmm(dy.double().abs(), w.double().abs()) + res.double().abs() + 1e-30
    e_y, e_dx = _rel(y, ref_y, mag_y), _rel(dx, ref_dx, mag_dx)
    f_y = _rel(torch.bmm(x, w.transpose(1, 2)), ref_y, mag_y)
    f_dx = _rel(torch.baddbmm(res, dy, w), ref_dx, mag_dx)
    assert e_y <= max(f_y, 1.2e-7) and e_dx <= max(f_dx, 1.2e-7), (e_y, f_y, e_dx, f_dx)
    assert e_y < 2e-7 and e_dx < 2e-7
    sums = st.view(nb, 2, cout)
    assert torch.allclose(sums[:, 0], ref_y.sum(1), rtol=1e-6, atol=1e-3)
    assert torch.allclose(sums[:, 1], (ref_y * ref_y).sum(1), rtol=1e-5)
    assert float(((y_ref.double() - y.double()).abs() / mag_y).max()) < 3e-7          # the other body: the same values
    assert float(((dx_ref.double() - dx.double()).abs() / mag_dx).max()) < 3e-7


@pytest.mark.parametrize("R,C", [(10368, 320), (2560, 320), (1500, 80), (777, 48), (4096, 640)])
def test_grouped_projection_on_the_xm_operand_buffer(R, C):
    """The grouped projection behind the aggregation (reference torch_vertex.py:57-61 + torch_nn.py:61) on the XM operand buffer
    (R, 2C) = [x chunk | m chunk] per conv group (include/gkg_hip.h "XM layout"): planes built with kperm, the activation read with
    row pitch 2C / batch stride C/2, the input gradient written in the same layout, the weight gradient added in the REFERENCE's
    interleaved column order — against fp64 on the reference's interleaved operand."""
    from gkgnet_amd import _lib
    from util import xm_interleaved
    lib = _lib.load()
    nb, ci, co = 4, C // 2, C // 2
    gen = torch.Generator(device="cuda").manual_seed(R + C)
    XM = torch.randn(R, 2 * C, device="cuda", generator=gen)
    w = torch.randn(nb, co, ci, device="cuda", generator=gen) * 0.1           # the reference's layout: columns x_0, m_0, x_1, m_1, ...
    dy = torch.randn(nb, R, co, device="cuda", generator=gen)
    pf, pd = _planes(lib, w, nb, co, ci, kperm=1)
    Ui = xm_interleaved(XM).view(R, nb, ci).permute(1, 0, 2).double()          # (nb, R, ci) in the reference's column order
    y = torch.full((nb, R, co), float("nan"), device="cuda")
    _lib.check(lib.gkg_linear_bn_fwd_x6(XM.data_ptr(), 2 * C, ci, pf.data_ptr(), y.data_ptr(), R, ci, co, nb, 0,
                                        *([None] * 10), 0.0, 0.0, None, None), "fwd")
    want_y = torch.bmm(Ui, w.double().transpose(1, 2))
    mag = torch.bmm(Ui.abs(), w.double().abs().transpose(1, 2)) + 1e-30
    assert _rel(y, want_y, mag) < 2e-7
    dXM = torch.full((R, 2 * C), float("nan"), device="cuda")
    _lib.check(lib.gkg_linear_dgrad_x6_sk(dy.data_ptr(), co, R * co, pd.data_ptr(), dXM.data_ptr(), R, ci, co, nb, None, None, 0,
                                          2 * C, ci, 0, None), "dgrad")
    want_dU = torch.bmm(dy.double(), w.double())                                # (nb, R, ci) interleaved columns
    got_dU = xm_interleaved(dXM).view(R, nb, ci).permute(1, 0, 2)
    magd = torch.bmm(dy.double().abs(), w.double().abs()) + 1e-30
    assert _rel(got_dU, want_dU, magd) < 2e-7
    dw = torch.zeros(nb, co, ci, device="cuda")
    _lib.check(lib.gkg_linear_wgrad_x6(dy.data_ptr(), co, R * co, XM.data_ptr(), 2 * C, ci, dw.data_ptr(), R, ci, co, nb, 1, None), "wgrad")
    want_dw = torch.bmm(dy.double().transpose(1, 2), Ui)
    magw = torch.bmm(dy.double().abs().transpose(1, 2), Ui.abs()) + 1e-30
    assert _rel(dw, want_dw, magw) < 2e-7
    # the batched launch takes the same problem (kperm in the descriptor)
    dw2 = torch.zeros(nb, co, ci, device="cuda")
    pr = (_lib.WgradProblem * 1)(_lib.WgradProblem(dy.data_ptr(), XM.data_ptr(), dw2.data_ptr(), R * co, ci, co, 2 * C, R, ci, co, nb, 1))
    _lib.check(lib.gkg_linear_wgrad_x6_batch(pr, 1, 0, None), "wgrad batch")
    assert _rel(dw2, want_dw, magw) < 2e-7
