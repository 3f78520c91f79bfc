"""Caches derived from parameters (x6 weight planes, bf16 / BN-folded weight copies) must notice EVERY optimiser step.
torch.optim's fused=True kernels update the parameters without moving their version counters (first assertion below), so the
staleness token is (version counter, process-wide optimiser-step counter) — gkgnet_amd/planes.py."""
import pytest
import torch


def _fused_adamw(params, lr):
    try:
        return torch.optim.AdamW(params, lr=lr, fused=True)
    except RuntimeError as e:                     # a build without fused kernels for this device
        pytest.skip(str(e))


def test_token_moves_with_a_fused_step_and_with_the_manual_mark():
    from gkgnet_amd.planes import mark_parameters_updated, param_version
    p = torch.nn.Parameter(torch.randn(16))
    p.grad = torch.randn(16)
    opt = _fused_adamw([p], 1e-2)
    t0, v0 = param_version(p), p._version
    opt.step()
    assert p._version == v0, "fused AdamW now moves the version counter: the step counter is no longer what catches it"
    t1 = param_version(p)
    assert t1 != t0
    with torch.no_grad():
        p.data.mul_(2.0)                          # an update through .data: invisible to both counters ...
    assert param_version(p) == t1
    mark_parameters_updated()                     # ... which is what this call is for
    assert param_version(p) != t1


def test_bf16_weight_copy_follows_a_fused_step():
    from gkgnet_amd import fused
    conv = torch.nn.Conv2d(8, 8, 1)
    w0 = fused._w16_of(conv).clone()
    assert fused._w16_of(conv) is fused._w16_of(conv)             # cached
    conv.weight.grad = torch.ones_like(conv.weight)
    _fused_adamw([conv.weight], 0.5).step()
    w1 = fused._w16_of(conv)
    assert not torch.equal(w0, w1)
    assert torch.equal(w1, conv.weight.detach().to(torch.bfloat16))


@pytest.mark.gpu
def test_x6_projections_follow_a_fused_step():
    """A Grapher block in training mode at 10 368 rows (its projections run on the split-bf16 kernels, weights read from the
    cached planes): forward, fused AdamW step, forward — the second output must equal the one computed after forcing a
    re-split of every weight, and differ from the first."""
    from gkgnet_amd import fused, layers
    from gkgnet_amd.grapher import Grapher
    layers.norm_cfg["type"] = "BN"
    B, C, H = 8, 64, 36
    torch.manual_seed(11)
    g = Grapher(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, drop_path=0.0, relative_pos=True,
                use_multi_group=True, num_group=4).cuda().train()
    x = torch.randn(B, C, H, H, device="cuda")
    opt = _fused_adamw(list(g.parameters()), 1e-2)
    y0 = g(x)
    y0.square().mean().backward()
    opt.step()
    with torch.no_grad():
        y1 = g(x)
        fused.refresh_weight_planes()
        y2 = g(x)
    assert not torch.equal(y0.detach(), y1)
    assert torch.equal(y1, y2), float((y1 - y2).abs().max())
