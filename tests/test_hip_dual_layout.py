"""A Grapher's NCHW output with a token-major companion for the GrapherLabel behind it (round 5, fused.DUAL_LAYOUT): same
values and gradients as the layout-pass form (reference torch_vertex.py:325-333 -> :392-403), engaged adaptively."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(dual: bool, steps: int, monkeypatch, record=None):
    from gkgnet_amd import fused
    from gkgnet_amd.grapher import Grapher, GrapherLabel
    monkeypatch.setattr(fused, "DUAL_LAYOUT", dual)
    torch.manual_seed(11)
    C, H, L, B = 64, 12, 20, 4
    g = Grapher(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=True, use_multi_group=True,
                num_group=2).cuda().train()
    gl = GrapherLabel(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=False, num_nodes=L,
                      use_multi_group=True, num_group=2).cuda().train()
    x = torch.randn(B, C, H, H, device="cuda").requires_grad_(True)
    e = torch.randn(B, L, C, device="cuda").requires_grad_(True)
    cx, ce = torch.randn(B, C, H, H, device="cuda"), torch.randn(B, L, C, device="cuda")
    outs = []
    for _ in range(steps):
        x.grad = e.grad = None
        g.zero_grad(set_to_none=True)
        gl.zero_grad(set_to_none=True)
        out = g(x)
        if record is not None:
            record.append(hasattr(out, "_gkg_tm"))
        e2, edge = gl(e, out)
        torch.autograd.backward([out, e2], [cx, ce])
        outs.append((out.detach().clone(), e2.detach().clone(), edge.clone(), x.grad.clone(), e.grad.clone(),
                     [p.grad.clone() for p in list(g.parameters()) + list(gl.parameters()) if p.grad is not None]))
    return outs


def test_companion_is_emitted_from_the_second_call_and_changes_nothing(monkeypatch):
    rec = []
    a = _run(True, 3, monkeypatch, rec)
    assert rec == [False, True, True]                     # the label branch asked for it during the first call
    b = _run(False, 3, monkeypatch)
    for step in (0, 1, 2):
        oa, ea, ia, gxa, gea, pa = a[step]
        ob, eb, ib, gxb, geb, pb = b[step]
        assert torch.equal(ia, ib)                                       # same graphs
        assert torch.allclose(oa, ob, atol=1e-5, rtol=1e-5) and torch.allclose(ea, eb, atol=1e-5, rtol=1e-5)
        # (conv biases in front of train-mode BN have an exactly zero gradient: what arrives there is rounding noise of ~1e-5
        # on both sides — hence the absolute floor)
        for u, v in [(gxa, gxb), (gea, geb)] + list(zip(pa, pb)):
            assert float((u - v).abs().max()) <= 2e-5 * float(v.abs().max()) + 5e-5, float((u - v).abs().max())


def test_only_one_of_the_two_outputs_used(monkeypatch):
    """The NCHW output alone (no label branch in this step) and the companion alone: the node's backward handles a missing
    upstream gradient on either side."""
    from gkgnet_amd import fused
    from gkgnet_amd.grapher import Grapher
    monkeypatch.setattr(fused, "DUAL_LAYOUT", True)
    torch.manual_seed(2)
    C, H, B = 32, 8, 2
    g = Grapher(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=True, use_multi_group=True,
                num_group=2).cuda().train()
    x = torch.randn(B, C, H, H, device="cuda").requires_grad_(True)
    cot = torch.randn(B, C, H, H, device="cuda")
    g(x).backward(cot)
    ref = x.grad.clone()
    g._gkg_want_tm = True
    x.grad = None
    out = g(x)
    assert hasattr(out, "_gkg_tm")
    out.backward(cot)
    assert torch.allclose(x.grad, ref, atol=1e-5, rtol=1e-4)
    x.grad = None
    out = g(x)
    tm = out._gkg_tm[1]
    assert torch.allclose(tm.view(B, H * H, C).transpose(1, 2).reshape(B, C, H, H), out, atol=0, rtol=0)
    tm.backward(cot.flatten(2).transpose(1, 2).reshape(-1, C))
    assert torch.allclose(x.grad, ref, atol=1e-5, rtol=1e-4)
