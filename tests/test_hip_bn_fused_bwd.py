"""The one-launch BN backward (round 5: statistics -> grid barrier -> apply from registers, csrc/gkg_dense.hip
bn_bwd_fused_kernel) through gkg_bn_bwd_atomic: against an fp64 evaluation of the BN (+ GELU) backward
(reference torch_nn.py:62-67 norm + act behind the 1x1 convolutions) and against the two-launch form it replaces."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(g, y, gamma, beta, eps, act):
    """fp64: dy, dgamma, dbeta of out = act(BN_train(y)) for upstream gradient g; y, g (nb, R, C)."""
    g, y, gamma, beta = g.double(), y.double(), gamma.double(), beta.double()
    mean = y.mean(1, keepdim=True)
    var = y.var(1, unbiased=False, keepdim=True)
    inv = (var + eps).rsqrt()
    yh = (y - mean) * inv
    z = yh * gamma[:, None, :] + beta[:, None, :]
    if act:
        cdf = 0.5 * (1 + torch.erf(z / 2 ** 0.5))
        dz = g * (cdf + z * torch.exp(-z * z / 2) / (2 * np.pi) ** 0.5)
    else:
        dz = g
    dbeta = dz.sum(1)
    dgamma = (dz * yh).sum(1)
    R = y.shape[1]
    dy = gamma[:, None, :] * inv * (dz - dbeta[:, None, :] / R - yh * dgamma[:, None, :] / R)
    return dy, dgamma, dbeta, mean[:, 0], inv[:, 0]


# (R, C, nb, act, scaled): the cfg2 shapes that take the one-launch form, ragged rows / channels below a tile, grouped
SHAPES = [(10368, 320, 1, 0, False), (2560, 1280, 1, 1, False), (2560, 160, 4, 1, False), (2560, 320, 1, 0, True),
          (777, 36, 1, 1, False), (130, 8, 3, 0, False), (4000, 640, 1, 1, True), (128, 64, 1, 0, False)]


@pytest.mark.parametrize("R,C,nb,act,scaled", SHAPES)
def test_one_launch_backward_matches_fp64_and_the_two_launch_form(R, C, nb, act, scaled):
    from gkgnet_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(R + C)
    y = torch.randn(nb, R, C, device="cuda", generator=gen) * 1.5 + 0.3
    g = torch.randn(nb, R, C, device="cuda", generator=gen)
    gamma = torch.rand(nb, C, device="cuda", generator=gen) + 0.5
    beta = torch.randn(nb, C, device="cuda", generator=gen) * 0.1
    eps = 1e-5
    rps = 7 if scaled else 0
    scale = (torch.rand((R + 6) // 7, device="cuda", generator=gen) + 0.5) if scaled else None
    g_eff = g if not scaled else g * scale.repeat_interleave(7)[:R].view(1, R, 1)
    want_dy, want_dg, want_db, mean, inv = _ref(g_eff, y, gamma, beta, eps, act)
    mean32, inv32 = mean.float().contiguous(), inv.float().contiguous()
    a = (gamma.double() * inv).float().contiguous()
    c = (beta.double() - gamma.double() * inv * mean).float().contiguous()
    res = []
    for flags in (2, 0):                                   # one launch (opt-in, up to 3 workgroups per CU) / two launches
        lib.gkg_bn_set_flags(flags)
        sums = torch.zeros(2, 2 * 4096 * 4, dtype=torch.float64, device="cuda")
        sums[1, :64] = 3.0                                  # the "other" buffer's dirty region: must come back zero
        dy = torch.full((nb, R, C), float("nan"), device="cuda")
        dg = torch.full((nb, C), float("nan"), device="cuda")
        db = torch.full((nb, C), float("nan"), device="cuda")
        if scaled:
            rc = lib.gkg_bn_bwd_atomic_scaled(g.data_ptr(), y.data_ptr(), a.data_ptr(), c.data_ptr(), mean32.data_ptr(),
                                              inv32.data_ptr(), dy.data_ptr(), dg.data_ptr(), db.data_ptr(), R, C, nb, C, R * C, act,
                                              sums[0].data_ptr(), sums[1].data_ptr(), 64, scale.data_ptr(), rps, None)
        else:
            rc = lib.gkg_bn_bwd_atomic(g.data_ptr(), y.data_ptr(), a.data_ptr(), c.data_ptr(), mean32.data_ptr(), inv32.data_ptr(),
                                       dy.data_ptr(), dg.data_ptr(), db.data_ptr(), R, C, nb, C, R * C, act, sums[0].data_ptr(),
                                       sums[1].data_ptr(), 64, None)
        _lib.check(rc, "gkg_bn_bwd_atomic")
        torch.cuda.synchronize()
        assert float(sums[1].abs().max()) == 0.0
        res.append((dy, dg, db))
    lib.gkg_bn_set_flags(0)
    assert lib.gkg_debug_barrier_timeouts() == 0
    for dy, dg, db in res:
        sc = float(want_dy.abs().max())
        assert float((dy.double() - want_dy).abs().max()) <= 2e-5 * sc
        assert torch.allclose(dg.double(), want_dg, rtol=2e-4, atol=2e-4 * float(want_dg.abs().max()))
        assert torch.allclose(db.double(), want_db, rtol=2e-4, atol=2e-4 * float(want_db.abs().max()))
    (d1, g1, b1), (d2, g2, b2) = res
    assert float((d1 - d2).abs().max()) <= 1e-5 * float(d2.abs().max())


def test_repeated_launches_and_capture_replay_keep_the_barrier_sound():
    """Hundreds of back-to-back barrier episodes (the generation word keeps counting, `arrive` is re-armed every time), eagerly and
    from a replayed hipGraph: no timeout, identical results."""
    from gkgnet_amd import _lib
    lib = _lib.load()
    lib.gkg_bn_set_flags(2)                                 # the one-launch form at 405 workgroups
    R, C = 10368, 320
    torch.manual_seed(0)
    y = torch.randn(1, R, C, device="cuda")
    g = torch.randn(1, R, C, device="cuda")
    a = torch.rand(C, device="cuda") + 0.5
    c = torch.randn(C, device="cuda")
    mean, inv = y.mean(1).view(-1).contiguous(), (y.var(1, unbiased=False) + 1e-5).rsqrt().view(-1).contiguous()
    sums = torch.zeros(2, 2 * 4096 * 4, dtype=torch.float64, device="cuda")
    dy = torch.empty_like(y)
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")

    def call(i):
        cur, other = sums[i & 1], sums[(i & 1) ^ 1]
        _lib.check(lib.gkg_bn_bwd_atomic(g.data_ptr(), y.data_ptr(), a.data_ptr(), c.data_ptr(), mean.data_ptr(), inv.data_ptr(),
                                         dy.data_ptr(), dg.data_ptr(), db.data_ptr(), R, C, 1, C, R * C, 1, cur.data_ptr(),
                                         other.data_ptr(), 2 * C, torch.cuda.current_stream().cuda_stream), "bn_bwd")
    for i in range(200):
        call(i)
    torch.cuda.synchronize()
    first = dy.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        call(0); call(1)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(8):
            call(i)
    for _ in range(20):
        gr.replay()
    torch.cuda.synchronize()
    lib.gkg_bn_set_flags(0)
    assert lib.gkg_debug_barrier_timeouts() == 0
    assert float((dy - first).abs().max()) <= 1e-5 * float(first.abs().max())
