"""BN backward statistics in the epilogue of the next projection's input-gradient GEMM (round 4: csrc/gkg_gemm_x6.hip X6_BNBWD,
fused._BnLink): in  h = GELU(BN(Y)) -> out = h W^T  the dgrad of the second projection produces the first layer's upstream
gradient, and its epilogue accumulates  sum dz, sum dz * yhat  from the accumulators and one read of Y — the first layer's
backward then runs its apply pass only.  Checked: the path is taken, and outputs / every gradient equal the two-pass form's
(the un-fused passes are pinned to the reference's fixtures F1-F7) — FFN (reference gkgnet.py:66-72: un-grouped producer) and
Grapher (torch_vertex.py:329-330: the grouped BasicConv in front of fc2)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _count_links(fused, monkeypatch):
    calls = {"n": 0}
    real = fused._dgrad_x6_with_link

    def counting(*a, **k):
        calls["n"] += 1
        return real(*a, **k)
    monkeypatch.setattr(fused, "_dgrad_x6_with_link", counting)
    return calls


def _close(a, b, tol=2e-4):
    return float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


def test_ffn_block_takes_the_epilogue_and_matches_the_two_pass_backward(monkeypatch):
    from gkgnet_amd import fused, layers
    from gkgnet_amd.backbone import FFN
    layers.norm_cfg["type"] = "BN"
    calls = _count_links(fused, monkeypatch)
    B, C, H = 2, 64, 72                                   # T = 10 368 rows: the fc2 dgrad runs on the x6 kernel
    xin = torch.randn(B, C, H, H, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    cot = torch.randn(B, C, H, H, device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    res = []
    for on in (True, False):
        monkeypatch.setattr(fused, "BN_EPILOGUE", on)
        monkeypatch.setattr(fused, "BN_EPILOGUE_MIN_ROWS", 0)
        torch.manual_seed(4)
        ffn = FFN(C, 4 * C, act="gelu").cuda().train()
        x = xin.clone().requires_grad_(True)
        before = calls["n"]
        out = ffn(x)
        out.backward(cot)
        assert (calls["n"] - before) == (1 if on else 0)
        res.append((out.detach(), x.grad, [p.grad.clone() for p in ffn.parameters() if p.grad is not None]))
    (o1, g1, p1), (o2, g2, p2) = res
    assert torch.equal(o1, o2)
    assert _close(g1, g2), float((g1 - g2).abs().max())
    assert len(p1) == len(p2) and all(_close(a, b) for a, b in zip(p1, p2))


def test_grapher_block_grouped_producer_and_label_branch(monkeypatch):
    from gkgnet_amd import fused, layers
    from gkgnet_amd.grapher import Grapher
    layers.norm_cfg["type"] = "BN"
    calls = _count_links(fused, monkeypatch)
    B, C, H, G = 8, 64, 36, 4                             # T = 10 368
    xin = torch.randn(B, C, H, H, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    cot = torch.randn(B, C, H, H, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    res = []
    for on in (True, False):
        monkeypatch.setattr(fused, "BN_EPILOGUE", on)
        monkeypatch.setattr(fused, "BN_EPILOGUE_MIN_ROWS", 0)
        torch.manual_seed(6)
        g = Grapher(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, drop_path=0.0, relative_pos=True,
                    use_multi_group=True, num_group=G).cuda().train()
        x = xin.clone().requires_grad_(True)
        before = calls["n"]
        out = g(x)
        out.backward(cot)
        assert (calls["n"] - before) == (1 if on else 0)       # fc2's dgrad carries the grouped BasicConv's statistics
        res.append((out.detach(), x.grad, [p.grad.clone() for p in g.parameters() if p.grad is not None]))
    (o1, g1, p1), (o2, g2, p2) = res
    assert torch.equal(o1, o2)
    assert _close(g1, g2), float((g1 - g2).abs().max())
    assert len(p1) == len(p2) and all(_close(a, b) for a, b in zip(p1, p2))


def test_a_second_consumer_of_the_activation_falls_back_cleanly(monkeypatch):
    """If autograd adds another contribution to the gradient the sums were taken from, the producer must notice (the tensor it
    receives is not the one the epilogue saw), discard the sums and run its own statistics pass: gradients still right, and
    the next layers' scratch protocol intact."""
    from gkgnet_amd import fused, layers
    layers.norm_cfg["type"] = "BN"
    monkeypatch.setattr(fused, "BN_EPILOGUE", True)
    monkeypatch.setattr(fused, "BN_EPILOGUE_MIN_ROWS", 0)
    R, C = 10368, 64
    gen = torch.Generator(device="cuda").manual_seed(8)
    x0 = torch.randn(R, C, device="cuda", generator=gen)
    res = []
    # "before": the second use of h is created BEFORE the x6 consumer, so the consumer's dx arrives first and autograd would add
    # the other gradient into it IN PLACE (same data_ptr, other values) unless the link holds the tensor
    for extra in ("after", "before", False):
        torch.manual_seed(9)
        s1 = torch.nn.Sequential(torch.nn.Conv2d(C, C, 1), layers.build_norm(C)).cuda().train()
        s2 = torch.nn.Sequential(torch.nn.Conv2d(C, C, 1), layers.build_norm(C)).cuda().train()
        x = x0.clone().requires_grad_(True)
        h = fused._lin(x, s1, act=1)
        side = (h * 0.5).sum() if extra == "before" else 0
        out = fused._lin(h, s2)
        if extra == "after":
            side = (h * 0.5).sum()                                           # a second use of h: autograd sums two gradients
        loss = out.square().sum() + side
        loss.backward()
        # reference: the same with the epilogue off
        monkeypatch.setattr(fused, "BN_EPILOGUE", False)
        torch.manual_seed(9)
        t1 = torch.nn.Sequential(torch.nn.Conv2d(C, C, 1), layers.build_norm(C)).cuda().train()
        t2 = torch.nn.Sequential(torch.nn.Conv2d(C, C, 1), layers.build_norm(C)).cuda().train()
        xr = x0.clone().requires_grad_(True)
        hr = fused._lin(xr, t1, act=1)
        outr = fused._lin(hr, t2)
        (outr.square().sum() + ((hr * 0.5).sum() if extra else 0)).backward()
        monkeypatch.setattr(fused, "BN_EPILOGUE", True)
        assert _close(x.grad, xr.grad, 5e-4), (extra, float((x.grad - xr.grad).abs().max()))
        for a, b in zip(s1.parameters(), t1.parameters()):
            if a.grad is not None:
                assert _close(a.grad, b.grad, 5e-4), extra
