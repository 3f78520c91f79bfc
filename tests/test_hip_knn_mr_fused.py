"""Row g2: the fused k-NN + max-relative aggregation kernel (gkg_knn_mr_fwd_tm, csrc/gkg_knn_tile.h MRF) through the C ABI.
Bit-exact against the C oracle (O.knn -> O.mr_fwd, the contract of reference torch_edge.py:164-176 -> torch_vertex.py:49-61)
and against the two-launch form it replaces (gkg_knn_fwd_tm -> gkg_mr_fwd_tm), incl. exact ties through the fused epilogue."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tm(a, B, G):
    """(B*G, c, T) channel-major groups -> (B, T, G*c) token-major."""
    BG, c, T = a.shape
    return np.ascontiguousarray(a.reshape(B, G, c, T).transpose(0, 3, 1, 2).reshape(B, T, G * c))


def _expect_U(x_tm, m_tm):
    """The grouped projection's operand buffer XM (B*N, 2C) from token-major x and m (include/gkg_hip.h "XM layout")."""
    B, N, C = x_tm.shape
    h = C // 4
    xf, mf = x_tm.reshape(B * N, 4, 1, h), m_tm.reshape(B * N, 4, 1, h)
    return np.ascontiguousarray(np.concatenate([xf, mf], axis=2).reshape(B * N, 2 * C)).astype(np.float32)


def _fused(x_tm, y_tm, rp, B, G, c, N, M, k, d, normalize=True, select=0):
    from gkgnet_amd import _lib
    lib = _lib.load()
    xd = torch.from_numpy(x_tm).cuda()
    yd = None if y_tm is None else torch.from_numpy(y_tm).cuda()
    rpd = None if rp is None else torch.from_numpy(rp).cuda()
    C = G * c
    flags = (_lib.KNN_NORMALIZE if normalize else 0) | select
    assert lib.gkg_knn_mr_fused_supported(B, G, c, N, M, k, d, 0 if y_tm is None else 1, 0 if rp is None else 1, flags) == 1
    U = torch.full((B * N, 2 * C), float("nan"), device="cuda")
    arg = torch.full((B, N, C), -1, dtype=torch.int16, device="cuda")
    nn16 = torch.full((B * G, N, k), -1, dtype=torch.int16, device="cuda")
    ws = torch.empty(lib.gkg_knn_workspace_bytes(B * G, c, N, M, k, d, _lib.F32, _lib.KNN_NORMALIZE), dtype=torch.uint8, device="cuda")
    e64 = torch.full((2, B * G, N, k), -1, dtype=torch.int64, device="cuda")
    yp, rpp = None if yd is None else yd.data_ptr(), None if rpd is None else rpd.data_ptr()
    # (a) x a plain (B, N, C) matrix: the kernel writes both halves of the operand buffer
    _lib.check(lib.gkg_knn_mr_fwd_tm(xd.data_ptr(), C, 0, yp, rpp, U.data_ptr(), arg.data_ptr(), nn16.data_ptr(), e64[0].data_ptr(),
                                     e64[1].data_ptr(), B, G, c, N, M, k, d, flags, ws.data_ptr(), ws.numel(), None), "gkg_knn_mr_fwd_tm")
    # (b) x lives in the buffer's x half already (what the Grapher's fc1 leaves): the queries, the centre rows and — self graph —
    # the neighbour rows are read through that view, only the m half is written.  Same bits.
    Ub = torch.full((B * N, 2 * C), float("nan"), device="cuda")
    Ub.view(B * N, 4, 2, C // 4)[:, :, 0] = xd.view(B * N, 4, C // 4)
    argb, nn16b = torch.full_like(arg, -1), torch.full_like(nn16, -1)
    _lib.check(lib.gkg_knn_mr_fwd_tm(Ub.data_ptr(), 2 * C, C // 4, yp, rpp, Ub.data_ptr(), argb.data_ptr(), nn16b.data_ptr(), None, None,
                                     B, G, c, N, M, k, d, flags, ws.data_ptr(), ws.numel(), None), "gkg_knn_mr_fwd_tm (in place)")
    assert torch.equal(Ub.view(torch.int32), U.view(torch.int32)) and torch.equal(argb, arg) and torch.equal(nn16b, nn16)
    # the two-launch form on the same inputs
    edge = torch.empty((2, B * G, N, k), dtype=torch.int64, device="cuda")
    _lib.check(lib.gkg_knn_fwd_tm(xd.data_ptr(), C, 0, yp, rpp, edge[0].data_ptr(), edge[1].data_ptr(), B, G, c, N, M, k, d, _lib.F32,
                                  flags, ws.data_ptr(), ws.numel(), None), "gkg_knn_fwd_tm")
    U2 = torch.empty_like(U)
    arg2 = torch.empty_like(arg)
    _lib.check(lib.gkg_mr_fwd_tm(xd.data_ptr(), C, 0, yp, edge[0].data_ptr(), U2.data_ptr(),
                                 arg2.data_ptr(), B, G, c, N, M, k, 1, _lib.F32, 1, None), "gkg_mr_fwd_tm")
    torch.cuda.synchronize()
    assert torch.equal(e64, edge)                                                # the optional int64 outputs: gkg_knn_fwd_tm's
    return (U.cpu().numpy(), arg.cpu().numpy().view(np.uint16), nn16.cpu().numpy().view(np.uint16),
            U2.cpu().numpy(), arg2.cpu().numpy().view(np.uint16), edge[0].cpu().numpy())


# (B, G, c, N, M(None = self), k, d, relpos): cfg2's Grapher and label graphs, cfg2ref (d = 3, 27-entry list, buffered form),
# ragged query tiles, k = 18 (the 18-slot gather), c = 20 (5 float4 columns), G = 1, small M
SHAPES = [(12, 4, 80, 324, None, 9, 1, True), (33, 4, 80, 80, 324, 9, 1, False), (22, 2, 320, 324, None, 9, 3, True),
          (64, 2, 24, 70, 150, 9, 1, False), (32, 2, 40, 200, None, 18, 1, True), (64, 2, 16, 100, 130, 4, 3, True),
          (32, 4, 20, 65, None, 5, 2, False), (43, 2, 32, 130, 40, 18, 2, False)]


@pytest.mark.parametrize("shape", SHAPES)
def test_fused_kernel_bit_exact_vs_oracle_and_two_launch_form(shape):
    from oracle import c_oracle as O
    B, G, c, N, M, k, d, relpos = shape
    rng = np.random.RandomState(N * 7 + c)
    x = rng.standard_normal((B * G, c, N)).astype(np.float32)
    y = None if M is None else rng.standard_normal((B * G, c, M)).astype(np.float32)
    Mk = N if M is None else M
    rp = -rng.random_sample((N, Mk)).astype(np.float32) if relpos else None
    want_idx, _ = O.knn(x, y, rp, k, d)
    want_m, _ = O.mr_fwd(x, y, want_idx)
    x_tm, y_tm = _tm(x, B, G), None if y is None else _tm(y, B, G)
    U, arg, nn16, U2, arg2, idx2 = _fused(x_tm, y_tm, rp, B, G, c, N, Mk, k, d)
    assert np.array_equal(nn16.astype(np.int64), want_idx)                       # the graph: bit-exact vs the oracle
    assert np.array_equal(idx2, want_idx)
    assert np.array_equal(U, _expect_U(x_tm, _tm(want_m, B, G)))                 # the aggregation: bit-exact vs the oracle
    assert np.array_equal(U, U2) and np.array_equal(arg, arg2)                   # and identical to the two-launch form
    # the winning rows are members of the query's list
    C = G * c
    a = arg.reshape(B, N, G, c)
    lists = want_idx.reshape(B, G, N, k).transpose(0, 2, 1, 3)                   # (B, N, G, k)
    assert (a[..., None] == lists[:, :, :, None, :]).any(-1).all()


@pytest.mark.parametrize("select", [0, 4, 8], ids=["auto", "direct", "buffered"])
def test_exact_ties_through_the_fused_epilogue(select):
    """Duplicated keys (exact distance ties: 'equal distance -> smaller key index first' decides membership, order and what
    survives the dilation) and duplicated VALUES (exact ties of the maximum: the first neighbour in list order wins)."""
    from oracle import c_oracle as O
    rng = np.random.RandomState(5)
    for B, G, c, N, M, k, d, use_rp in [(43, 2, 16, 130, None, 6, 2, True), (64, 2, 16, 90, 150, 9, 2, False),
                                        (64, 4, 8, 64, None, 9, 1, False)]:
        x = rng.standard_normal((B * G, c, N)).astype(np.float32)
        y = None if M is None else rng.standard_normal((B * G, c, M)).astype(np.float32)
        t = x if y is None else y
        T = t.shape[2] // 3
        t[:, :, T:2 * T] = t[:, :, :T]
        t[:, :, 2 * T:3 * T] = t[:, :, :T]
        Mk = N if M is None else M
        rp = (np.round(-rng.random_sample((N, Mk)) * 4) / 4).astype(np.float32) if use_rp else None
        want_idx, _ = O.knn(x, y, rp, k, d)
        want_m, _ = O.mr_fwd(x, y, want_idx)
        x_tm, y_tm = _tm(x, B, G), None if y is None else _tm(y, B, G)
        U, arg, nn16, U2, arg2, _ = _fused(x_tm, y_tm, rp, B, G, c, N, Mk, k, d, select=select)
        assert np.array_equal(nn16.astype(np.int64), want_idx), (B, G, c, N, M, k, d)
        assert np.array_equal(U, _expect_U(x_tm, _tm(want_m, B, G)))
        assert np.array_equal(U, U2) and np.array_equal(arg, arg2)


def test_non_finite_inputs_follow_the_two_launch_form():
    """NaN / inf tokens: the careful maximum chain (a NaN is the maximum and sticks) — same bits as gkg_mr_fwd_tm."""
    rng = np.random.RandomState(3)
    B, G, c, N, k = 64, 2, 16, 100, 9
    x = rng.standard_normal((B * G, c, N)).astype(np.float32)
    x[0, 3, 17] = np.nan
    x[1, 5, 40] = np.inf
    x[2, 0, 99] = -np.inf
    x_tm = _tm(x, B, G)
    U, arg, nn16, U2, arg2, idx2 = _fused(x_tm, None, None, B, G, c, N, N, k, 1)
    assert np.array_equal(nn16.astype(np.int64), idx2)
    assert np.array_equal(U.view(np.uint32), U2.view(np.uint32)) and np.array_equal(arg, arg2)


def test_unsupported_shapes_are_reported():
    from gkgnet_amd import _lib
    lib = _lib.load()
    f = _lib.KNN_NORMALIZE
    assert lib.gkg_knn_mr_fused_supported(32, 4, 80, 324, 324, 9, 1, 0, 1, f) == 1          # cfg2 Grapher graph
    assert lib.gkg_knn_mr_fused_supported(32, 4, 80, 80, 324, 9, 1, 1, 0, f) == 1           # cfg2 label graph
    assert lib.gkg_knn_mr_fused_supported(1, 1, 40, 80, 20736, 9, 1, 1, 0, f) == 0          # few queries, many keys: key splits
    assert lib.gkg_knn_mr_fused_supported(2, 1, 8, 50, 70, 32, 2, 1, 0, f) == 0             # 64-entry list
    assert lib.gkg_knn_mr_fused_supported(4, 2, 200, 1296, 1296, 9, 2, 0, 1, f | _lib.KNN_RELPOS_UNIT) == 0   # prefilter shape
    assert lib.gkg_knn_mr_fused_supported(2, 2, 18, 100, 100, 9, 1, 0, 0, f) == 0           # C % 16 != 0


def _block(knn_mr: bool, monkeypatch):
    from gkgnet_amd import block, fused
    from gkgnet_amd.grapher import Grapher, GrapherLabel
    monkeypatch.setattr(fused, "KNN_MR", knn_mr)
    monkeypatch.setattr(block, "ENABLED", False)     # this test counts the per-layer composition's calls (the block driver issues
                                                     # the same launches from C: tests/test_hip_block_driver.py holds it to these bits)
    torch.manual_seed(21)
    C, H, L, B = 64, 12, 20, 48                     # 96 problems x 3 / 1 query tiles: no key splits, the fused form applies
    g = Grapher(C, 9, 2, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=True, use_multi_group=True,
                num_group=2).cuda().train()
    gl = GrapherLabel(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=False, num_nodes=L,
                      use_multi_group=True, num_group=2).cuda().train()
    x = torch.randn(B, C, H, H, device="cuda").requires_grad_(True)
    e = torch.randn(B, L, C, device="cuda").requires_grad_(True)
    cx, ce = torch.randn(B, C, H, H, device="cuda"), torch.randn(B, L, C, device="cuda")
    calls = []
    real = fused._KnnMaxRelativeTM.forward
    monkeypatch.setattr(fused._KnnMaxRelativeTM, "forward", staticmethod(lambda *a: (calls.append(1), real(*a))[1]))
    out = g(x)
    e2, edge = gl(e, out)
    torch.autograd.backward([out, e2], [cx, ce])
    monkeypatch.setattr(fused._KnnMaxRelativeTM, "forward", staticmethod(real))
    return len(calls), (out.detach(), e2.detach(), edge.clone(), x.grad.clone(), e.grad.clone())


def test_blocks_take_the_fused_kernel_and_nothing_changes(monkeypatch):
    n1, a = _block(True, monkeypatch)
    n0, b = _block(False, monkeypatch)
    assert n1 == 2 and n0 == 0                       # Grapher graph + label graph on the fused kernel
    assert torch.equal(a[2], b[2])                   # GrapherLabel's returned edge_index: identical
    for u, v in zip(a, b):
        assert torch.equal(u, v)                     # same arithmetic in the same order: identical bits


def test_fuzz_fused_kernel_random_shapes():
    """Random eligible shapes (ragged tiles, odd channel quads, every list size up to 36, self and bipartite graphs, with and
    without a bias, exact duplicates): the fused kernel against the C oracle and the two-launch form, ~10 s."""
    import time
    from gkgnet_amd import _lib
    from oracle import c_oracle as O
    lib = _lib.load()
    rng = np.random.RandomState(123)
    t0, done = time.time(), 0
    while time.time() - t0 < 10.0:
        G = int(rng.choice([1, 2, 4]))
        c = int(rng.choice([4, 8, 12, 16, 20, 40, 80])) * (4 // G if G < 4 else 1)
        if (G * c) % 16:
            continue
        N = int(rng.randint(20, 400))
        self_graph = rng.rand() < 0.5
        M = N if self_graph else int(rng.randint(40, 600))
        d = int(rng.randint(1, 4))
        k = int(rng.choice([3, 5, 9, 12, 18]))
        if k * d > min(M, 36):
            continue
        use_rp = rng.rand() < 0.5
        qt = (N + 63) // 64
        B = max(1, (300 + qt * G - 1) // (qt * G))                # enough workgroups that the plan does not split the keys
        flags = _lib.KNN_NORMALIZE
        if not lib.gkg_knn_mr_fused_supported(B, G, c, N, M, k, d, 0 if self_graph else 1, 1 if use_rp else 0, flags):
            continue
        x = rng.standard_normal((B * G, c, N)).astype(np.float32)
        y = None if self_graph else rng.standard_normal((B * G, c, M)).astype(np.float32)
        if rng.rand() < 0.3:                                       # exact duplicates: ties decide membership and the maximum
            t = x if y is None else y
            T = t.shape[2] // 3
            if T:
                t[:, :, T:2 * T] = t[:, :, :T]
        rp = (np.round(-rng.random_sample((N, M)) * 8) / 8).astype(np.float32) if use_rp else None
        want_idx, _ = O.knn(x, y, rp, k, d)
        want_m, _ = O.mr_fwd(x, y, want_idx)
        x_tm, y_tm = _tm(x, B, G), None if y is None else _tm(y, B, G)
        U, arg, nn16, U2, arg2, idx2 = _fused(x_tm, y_tm, rp, B, G, c, N, M, k, d)
        tag = (B, G, c, N, M, k, d, use_rp)
        assert np.array_equal(nn16.astype(np.int64), want_idx), tag
        assert np.array_equal(U, _expect_U(x_tm, _tm(want_m, B, G))), tag
        assert np.array_equal(U, U2) and np.array_equal(arg, arg2), tag
        done += 1
    assert done >= 5, done
