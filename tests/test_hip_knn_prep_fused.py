"""Round 6: the Grapher's fc1 BN-apply and the k-NN's token preparation as ONE pass (gkg_bn_apply_knn_prep,
reference torch_vertex.py:326 -> torch_edge.py:167-173).  Through the C ABI: bit-identical x, saved statistics, graphs and
aggregation against gkg_bn_apply_train followed by the k-NN call's own preparation — self graph (fused k-NN + aggregation and the
compact two-launch form), pooled / bipartite keys, the prefilter form, an XM operand buffer and a plain output.  And at module
level: identical bits with the fusion on and off, and no token_prep launch for the queries."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(B, G, c, N, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    C = G * c
    Y = torch.randn(B * N, C, device="cuda", generator=g) * 1.7 + 0.3
    sums = torch.stack([Y.double().sum(0), (Y.double() ** 2).sum(0)]).contiguous()        # [2][C]: what the GEMM epilogue leaves
    gamma = torch.rand(C, device="cuda", generator=g) + 0.5
    beta = torch.randn(C, device="cuda", generator=g) * 0.1
    return Y, sums, gamma, beta


def _bn_outs(C):
    return [torch.full((C,), float("nan"), device="cuda") for _ in range(4)]


# (B, G, c, N, M (None = self), k, d, relpos, fused_mr, xm)
CASES = [(32, 4, 80, 324, None, 9, 1, True, 1, True),        # cfg2 Grapher graph: fused k-NN + aggregation, x in the XM buffer
         (32, 4, 80, 80, 324, 9, 1, False, 1, True),         # cfg2 label graph
         (6, 2, 40, 1300, 325, 9, 1, True, 0, True),         # pooled keys, two-launch form (u16 lists)
         (3, 2, 200, 1296, None, 9, 2, True, 0, True),       # stage-3 shape: the prefilter kernel (planes prepared too)
         (5, 4, 20, 200, None, 5, 1, False, 1, False),       # plain (T, C) output, narrow groups
         (2, 1, 64, 777, 300, 18, 2, True, 0, False)]


@pytest.mark.parametrize("case", CASES)
def test_prepared_queries_give_identical_bits(case):
    from gkgnet_amd import _lib
    lib = _lib.load()
    B, G, c, N, M, k, d, relpos, fused_mr, xm = case
    C = G * c
    Mk = N if M is None else M
    Y, sums, gamma, beta = _setup(B, G, c, N, 7 * N + c)
    g = torch.Generator(device="cuda").manual_seed(N)
    y = None if M is None else torch.randn(B, Mk, C, device="cuda", generator=g)
    rp = (-torch.rand(N, Mk, device="cuda", generator=g)) if relpos else None
    flags = _lib.KNN_NORMALIZE | (_lib.KNN_RELPOS_UNIT if relpos else 0)
    if fused_mr:
        assert lib.gkg_knn_mr_fused_supported(B, G, c, N, Mk, k, d, 0 if y is None else 1, 1 if relpos else 0, flags) == 1
    ld, chunk = (2 * C, C // 4) if xm else (C, 0)
    wsb = lib.gkg_knn_workspace_bytes(B * G, c, N, Mk, k, d, _lib.F32, _lib.KNN_NORMALIZE)
    yp, rpp = (None if y is None else y.data_ptr()), (None if rp is None else rp.data_ptr())

    def run(fusedprep):
        out = torch.full((B * N, ld), float("nan"), device="cuda")
        a, cs, mean, invstd = _bn_outs(C)
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        nbt = torch.zeros((), dtype=torch.int64, device="cuda")
        ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
        f = flags
        if fusedprep:
            _lib.check(lib.gkg_bn_apply_knn_prep(Y.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, rm.data_ptr(),
                                                 rv.data_ptr(), nbt.data_ptr(), a.data_ptr(), cs.data_ptr(), mean.data_ptr(),
                                                 invstd.data_ptr(), out.data_ptr(), ld, chunk, B, G, c, N, Mk, k, d,
                                                 0 if y is None else 1, 1 if relpos else 0, flags, fused_mr, 0, None, None,
                                                 ws.data_ptr(), wsb, 0.1, 1e-5, None, 0, None), "gkg_bn_apply_knn_prep")
            f |= _lib.KNN_X_PREPARED
        else:
            _lib.check(lib.gkg_bn_apply_train(Y.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, rm.data_ptr(),
                                              rv.data_ptr(), nbt.data_ptr(), a.data_ptr(), cs.data_ptr(), mean.data_ptr(),
                                              invstd.data_ptr(), None, out.data_ptr(), B * N, C, 1, ld, 0, chunk, 0, 0, None, 0,
                                              0.1, 1e-5, None, 0, None), "gkg_bn_apply_train")
        res = dict(a=a, c=cs, mean=mean, invstd=invstd, rm=rm, rv=rv, nbt=nbt)
        if fused_mr:
            XM = out if xm else torch.full((B * N, 2 * C), float("nan"), device="cuda")
            arg = torch.empty((B, N, C), dtype=torch.int16, device="cuda")
            nn16 = torch.empty((B * G, N, k), dtype=torch.int16, device="cuda")
            _lib.check(lib.gkg_knn_mr_fwd_tm(out.data_ptr(), ld, chunk, yp, rpp, XM.data_ptr(), arg.data_ptr(), nn16.data_ptr(), None, None,
                                             B, G, c, N, Mk, k, d, f, ws.data_ptr(), wsb, None), "gkg_knn_mr_fwd_tm")
            res.update(XM=XM, arg=arg, nn16=nn16)
        else:
            nn16 = torch.empty((B * G, N, k), dtype=torch.int16, device="cuda")
            _lib.check(lib.gkg_knn_fwd_tm16(out.data_ptr(), ld, chunk, yp, rpp, nn16.data_ptr(), B, G, c, N, Mk, k, d, _lib.F32, f,
                                            ws.data_ptr(), wsb, None), "gkg_knn_fwd_tm16")
            res.update(out=out, nn16=nn16)
        torch.cuda.synchronize()
        return res

    r0, r1 = run(False), run(True)
    for key in r0:
        t0, t1 = r0[key], r1[key]
        if t0.dtype == torch.float32:
            assert torch.equal(t0.view(torch.int32), t1.view(torch.int32)), (case, key)
        else:
            assert torch.equal(t0, t1), (case, key)
    assert int(r1["nbt"]) == 1
    # and the graph is the oracle's on the BN-applied tokens (ties aside: none in random data)
    from oracle import c_oracle as O
    x_tm = (r0["out"] if "out" in r0 else r0["XM"])
    if xm:
        x_tm = x_tm.view(B * N, 4, 2, C // 4)[:, :, 0].reshape(B, N, C)
    else:
        x_tm = x_tm.view(B, N, C) if "out" in r0 else None
    if x_tm is not None and B * G * N * Mk <= 3e7:
        cm = lambda t: np.ascontiguousarray(t.reshape(t.shape[0], t.shape[1], G, c).permute(0, 2, 3, 1).reshape(-1, c, t.shape[1]).cpu().numpy())
        want, _ = O.knn(cm(x_tm), None if y is None else cm(y), None if rp is None else rp.cpu().numpy(), k, d)
        assert np.array_equal(r1["nn16"].cpu().numpy().view(np.uint16).astype(np.int64), want), case


def _block(prep: bool, monkeypatch, r=1):
    from gkgnet_amd import fused, _lib
    from gkgnet_amd.grapher import Grapher, GrapherLabel
    monkeypatch.setattr(fused, "KNN_PREP", prep)
    torch.manual_seed(21)
    C, H, L, B = 64, 12, 20, 48
    g = Grapher(C, 9, 2, "mr", "gelu", "batch", True, False, 0.2, r, n=H * H, relative_pos=True, use_multi_group=True,
                num_group=2).cuda().train()
    gl = GrapherLabel(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=False, num_nodes=L,
                      use_multi_group=True, num_group=2).cuda().train()
    x = torch.randn(B, C, H, H, device="cuda").requires_grad_(True)
    e = torch.randn(B, L, C, device="cuda").requires_grad_(True)
    cx, ce = torch.randn(B, C, H, H, device="cuda"), torch.randn(B, L, C, device="cuda")
    _lib.prof_reset()
    _lib.prof_enable(True)
    out = g(x)
    e2, edge = gl(e, out)
    torch.autograd.backward([out, e2], [cx, ce])
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    launches = _lib.prof_read()["token_prep"][1]
    return launches, (out.detach(), e2.detach(), edge.clone(), x.grad.clone(), e.grad.clone(),
                      [p.grad.clone() for p in list(g.parameters()) + list(gl.parameters()) if p.grad is not None],
                      [b.clone() for b in list(g.buffers()) + list(gl.buffers())])


@pytest.mark.parametrize("r", [1, 2])
def test_blocks_identical_bits_and_fewer_preparation_launches(r, monkeypatch):
    n1, a = _block(True, monkeypatch, r)
    n0, b = _block(False, monkeypatch, r)
    # with the fusion: the Grapher's self graph needs no preparation launch at all (r == 1; pooled keys: theirs), the label graph
    # one for its keys only, and those two producer passes ARE the BN-apply launches; without: one per graph on top of the applies
    # (launches the library counts under token_prep; the two producer passes replace the two BN-apply launches of fc1)
    assert n0 == 2 and n1 == (3 if r == 1 else 4), (n0, n1)
    for u, v in zip(a[:5], b[:5]):
        assert torch.equal(u, v)
    for u, v in zip(a[5], b[5]):
        assert torch.allclose(u, v, rtol=1e-4, atol=1e-3)            # weight gradients: atomically accumulated (run-dependent order)
    for u, v in zip(a[6], b[6]):
        assert torch.equal(u, v)


def test_keys_producer_matches_apply_dual_plus_own_preparation():
    """as_keys: a Grapher's fc2 BN-apply (+ token-major residual, both output layouts) that also prepares the KEYS of the label graph
    behind it, against gkg_bn_apply_train_dual followed by the label call's own preparation — same bits everywhere."""
    from gkgnet_amd import _lib
    lib = _lib.load()
    for (B, G, c, L, M, k) in [(32, 4, 80, 80, 324, 9), (3, 2, 40, 80, 1296, 9), (2, 2, 32, 20, 5184, 9)]:
        C = G * c
        Y, sums, gamma, beta = _setup(B, G, c, M, 3 * M + c)
        g = torch.Generator(device="cuda").manual_seed(M)
        res = torch.randn(B * M, C, device="cuda", generator=g)
        xq = torch.randn(B, L, C, device="cuda", generator=g)                     # the label queries
        flags = _lib.KNN_NORMALIZE
        fused_mr = lib.gkg_knn_mr_fused_supported(B, G, c, L, M, k, 1, 1, 0, flags)
        wsb = lib.gkg_knn_workspace_bytes(B * G, c, L, M, k, 1, _lib.F32, _lib.KNN_NORMALIZE)

        def run(producer):
            nchw = torch.full((B, C, M), float("nan"), device="cuda")
            tm = torch.full((B * M, C), float("nan"), device="cuda")
            a, cs, mean, invstd = _bn_outs(C)
            ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
            f = flags
            if producer:
                _lib.check(lib.gkg_bn_apply_knn_prep(Y.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, None, None,
                                                     None, a.data_ptr(), cs.data_ptr(), mean.data_ptr(), invstd.data_ptr(), tm.data_ptr(),
                                                     0, 0, B, G, c, L, M, k, 1, 1, 0, flags, fused_mr, 1, res.data_ptr(), nchw.data_ptr(),
                                                     ws.data_ptr(), wsb, 0.1, 1e-5, None, 0, None), "gkg_bn_apply_knn_prep (keys)")
                f |= _lib.KNN_Y_PREPARED
            else:
                _lib.check(lib.gkg_bn_apply_train_dual(Y.data_ptr(), sums.data_ptr(), gamma.data_ptr(), beta.data_ptr(), None, None, None,
                                                       None, a.data_ptr(), cs.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                       res.data_ptr(), nchw.data_ptr(), tm.data_ptr(), B, C, M, 0.1, 1e-5, None, 0, None),
                           "gkg_bn_apply_train_dual")
            nn16 = torch.empty((B * G, L, k), dtype=torch.int16, device="cuda")
            _lib.check(lib.gkg_knn_fwd_tm16(xq.data_ptr(), 0, 0, tm.data_ptr(), None, nn16.data_ptr(), B, G, c, L, M, k, 1, _lib.F32, f,
                                            ws.data_ptr(), wsb, None), "gkg_knn_fwd_tm16")
            torch.cuda.synchronize()
            return dict(nchw=nchw, tm=tm, a=a, c=cs, mean=mean, invstd=invstd, nn16=nn16)

        r0, r1 = run(False), run(True)
        for key in r0:
            t0, t1 = r0[key], r1[key]
            same = torch.equal(t0.view(torch.int32), t1.view(torch.int32)) if t0.dtype == torch.float32 else torch.equal(t0, t1)
            assert same, (B, G, c, L, M, key)


def test_second_step_prepares_the_label_keys_in_the_grapher(monkeypatch):
    """Two steps of Grapher -> GrapherLabel: from the second step on (after the label block has told its producer which k-NN it
    solves) the Grapher's last pass prepares the label graph's keys, the label block's fc1 its queries — no stand-alone token
    preparation launch anywhere — and nothing changes in the results."""
    from gkgnet_amd import fused, _lib
    from gkgnet_amd.grapher import Grapher, GrapherLabel

    def two_steps(prep):
        monkeypatch.setattr(fused, "KNN_PREP", prep)
        torch.manual_seed(5)
        C, H, L, B = 64, 12, 20, 48
        g = Grapher(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=True, use_multi_group=True,
                    num_group=2).cuda().train()
        gl = GrapherLabel(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=False, num_nodes=L,
                          use_multi_group=True, num_group=2).cuda().train()
        gen = torch.Generator(device="cuda").manual_seed(9)
        outs = []
        for step in range(2):
            x = torch.randn(B, C, H, H, device="cuda", generator=gen).requires_grad_(True)
            e = torch.randn(B, L, C, device="cuda", generator=gen).requires_grad_(True)
            cx, ce = torch.randn(B, C, H, H, device="cuda", generator=gen), torch.randn(B, L, C, device="cuda", generator=gen)
            _lib.prof_reset()
            _lib.prof_enable(True)
            out = g(x)
            e2, edge = gl(e, out)
            torch.autograd.backward([out, e2], [cx, ce])
            torch.cuda.synchronize()
            _lib.prof_enable(False)
            outs.append((_lib.prof_read()["token_prep"][1], out.detach().clone(), e2.detach().clone(), edge.clone(), x.grad.clone(),
                         e.grad.clone()))
        return outs

    on, off = two_steps(True), two_steps(False)
    assert [o[0] for o in off] == [2, 2]
    # step 1: fc1 producer (Grapher), fc1 producer (label) + the label keys' own launch; step 2: three producers, each one a BN-apply
    assert [o[0] for o in on] == [3, 3]
    for a, b in zip(on, off):
        for u, v in zip(a[1:], b[1:]):
            assert torch.equal(u, v)
