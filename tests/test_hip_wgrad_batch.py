"""Batched weight gradients (round 5): gkg_linear_wgrad_x6_batch through the C ABI against fp64, and the deferred queue of
gkgnet_amd.fused (weight gradients of a backward pass issued as one launch at its end) against the per-layer launches.
Reference: the weight gradients of torch_vertex.py:290-306 (fc1 / fc2), :334-360 (FFNLabel), torch_nn.py:57-69 (BasicConv)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _problems(shapes, gen):
    from gkgnet_amd import _lib
    ts, ps = [], []
    for R, cin, cout, nb in shapes:
        dy = torch.randn(nb, R, cout, device="cuda", generator=gen)
        x = torch.randn(nb, R, cin, device="cuda", generator=gen) * 2
        dw = torch.zeros(nb, cout, cin, device="cuda")
        ts.append((dy, x, dw))
        ps.append(_lib.WgradProblem(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), R * cout, R * cin, cout, cin, R, cin, cout, nb))
    return ts, ps


def _check(ts):
    for dy, x, dw in ts:
        ref = torch.bmm(dy.double().transpose(1, 2), x.double())
        mag = torch.bmm(dy.double().abs().transpose(1, 2), x.double().abs()) + 1e-30
        e = float(((dw.double() - ref).abs() / mag).max())
        f = float(((torch.bmm(dy.transpose(1, 2), x).double() - ref).abs() / mag).max())
        assert e <= max(f, 1.2e-7) and e < 2e-7, (tuple(dy.shape), tuple(x.shape), e, f)


# the eight weight gradients of the cfg2 step (Grapher rows 10 368, label rows 2 560)
CFG2 = [(2560, 1280, 320, 1), (2560, 320, 1280, 1), (2560, 640, 320, 1), (2560, 160, 160, 4), (2560, 320, 320, 1),
        (10368, 640, 320, 1), (10368, 160, 160, 4), (10368, 320, 320, 1)]
# ragged rows (register-load remainder + its own launch), fewer than 12 / 24 / 48 units (1 / 2 / 4 slabs shared by XCDs), one
# unit, unaligned widths (falls out of the batch), three slabs' worth that does not divide 8
ODD = [(777, 36, 40, 1), (1280, 64, 64, 1), (128, 32, 64, 1), (3000, 1, 3, 1), (640, 256, 100, 2), (4100, 400, 100, 1),
       (5120, 80, 80, 1), (19, 16, 8, 1), (2048 + 128 * 9, 200, 72, 1), (129, 64, 8, 2)]


# GKGNet-576's stage widths (80 / 160 / 400, grouped 40): widths that pad to the 64 x 128 and 64 x 64 tile forms in one launch
STAGES = [(6400, 80, 80, 1), (6400, 160, 80, 1), (6400, 80, 320, 1), (6400, 320, 80, 1), (5120, 40, 40, 4), (3200, 160, 160, 1),
          (3200, 160, 640, 1), (1296, 400, 400, 1), (1296, 1600, 400, 1), (2560, 320, 320, 1)]


@pytest.mark.parametrize("shapes", [CFG2, ODD, CFG2 + ODD + CFG2[:3], STAGES], ids=["cfg2", "odd", "two_launches", "stage_widths"])
def test_batch_matches_fp64(shapes):
    from gkgnet_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device="cuda").manual_seed(len(shapes))
    ts, ps = _problems(shapes, gen)
    arr = (_lib.WgradProblem * len(ps))(*ps)
    _lib.check(lib.gkg_linear_wgrad_x6_batch(arr, len(ps), 0, None), "gkg_linear_wgrad_x6_batch")
    torch.cuda.synchronize()
    _check(ts)


def test_batch_rejects_bad_arguments():
    from gkgnet_amd import _lib
    lib = _lib.load()
    assert lib.gkg_linear_wgrad_x6_batch(None, 1, 0, None) != 0
    p = (_lib.WgradProblem * 1)(_lib.WgradProblem(None, None, None, 0, 0, 4, 4, 128, 4, 4, 1))
    assert lib.gkg_linear_wgrad_x6_batch(p, 1, 0, None) != 0
    assert b"null" in lib.gkg_last_error_string()


def _block_grads(batch: bool, graph: bool, monkeypatch):
    """Parameter gradients of one Grapher -> GrapherLabel fwd+bwd step with a GradBucket."""
    from gkgnet_amd import fused, parallel
    from gkgnet_amd.grapher import Grapher, GrapherLabel
    monkeypatch.setattr(fused, "WGRAD_BATCH", batch)
    torch.manual_seed(3)
    C, H, L, B = 64, 16, 24, 8                      # T = 2048 rows (whole units), label rows 192 (ragged: own launches)
    g = Grapher(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=True, use_multi_group=True,
                num_group=2).cuda().train()
    gl = GrapherLabel(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=False, num_nodes=L,
                      use_multi_group=True, num_group=2).cuda().train()
    params = list(g.parameters()) + list(gl.parameters())
    bucket = parallel.GradBucket(params)
    x = torch.randn(B, C, H, H, device="cuda").requires_grad_(True)
    e = torch.randn(B, L, C, device="cuda").requires_grad_(True)
    cx, ce = torch.randn(B, C, H, H, device="cuda"), torch.randn(B, L, C, device="cuda")

    def step():
        bucket.release(prezero=True)
        x.grad = None
        e.grad = None
        out = g(x)
        e2, _ = gl(e, out)
        torch.autograd.backward([out, e2], [cx, ce])
        bucket.pack()

    step()
    torch.cuda.synchronize()
    if graph:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            step()
        bucket.flat.fill_(float("nan"))
        gr.replay()
        gr.replay()
    torch.cuda.synchronize()
    assert not fused._WQ.items and not fused._WQ.keep
    return bucket.flat.clone(), x.grad.clone(), e.grad.clone()


@pytest.mark.parametrize("graph", [False, True], ids=["eager", "hipgraph"])
def test_deferred_queue_gives_the_per_layer_gradients(graph, monkeypatch):
    a = _block_grads(True, graph, monkeypatch)
    b = _block_grads(False, False, monkeypatch)
    for u, v in zip(a, b):
        assert torch.isfinite(u).all()
        scale = float(v.abs().max())
        assert float((u - v).abs().max()) <= 2e-5 * scale, float((u - v).abs().max()) / scale


def test_without_a_bucket_nothing_is_queued(monkeypatch):
    from gkgnet_amd import fused
    from gkgnet_amd.grapher import Grapher
    monkeypatch.setattr(fused, "WGRAD_BATCH", True)
    torch.manual_seed(0)
    g = Grapher(32, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=256, relative_pos=True, use_multi_group=True,
                num_group=2).cuda().train()
    x = torch.randn(8, 32, 16, 16, device="cuda").requires_grad_(True)
    g(x).sum().backward()
    assert not fused._WQ.items and not fused._WQ.keep
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in g.parameters() if p.requires_grad and p.dim() > 1)


def test_backward_on_a_side_stream_launches_the_batch_there(monkeypatch):
    """Forward and backward under a non-default stream, the engine callback fires in the calling thread: the batch must go to
    the stream its operands were produced on (gradients complete and correct after synchronising THAT stream only)."""
    from gkgnet_amd import fused, parallel
    from gkgnet_amd.grapher import Grapher
    monkeypatch.setattr(fused, "WGRAD_BATCH", True)
    torch.manual_seed(5)
    g = Grapher(64, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=256, relative_pos=True, use_multi_group=True,
                num_group=2).cuda().train()
    x = torch.randn(8, 64, 16, 16, device="cuda").requires_grad_(True)
    cot = torch.randn(8, 64, 16, 16, device="cuda")
    params = list(g.parameters())
    # reference gradients: per-layer launches on the default stream
    monkeypatch.setattr(fused, "WGRAD_BATCH", False)
    g(x).backward(cot)
    torch.cuda.synchronize()
    want = {n: p.grad.clone() for n, p in g.named_parameters() if p.grad is not None}
    g.zero_grad(set_to_none=True)
    x.grad = None
    monkeypatch.setattr(fused, "WGRAD_BATCH", True)
    bucket = parallel.GradBucket(params)
    bucket.release(prezero=True)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = g(x)
        out.backward(cot)
    s.synchronize()                                     # only the side stream
    got = {n: p.grad for n, p in g.named_parameters() if p.grad is not None}
    for n, v in want.items():
        assert float((got[n] - v).abs().max()) <= 2e-5 * float(v.abs().max()) + 5e-5, n
    torch.cuda.synchronize()
