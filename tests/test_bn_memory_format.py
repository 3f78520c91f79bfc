"""The PyTorch behaviour DESIGN.md §2 describes: CPU ``batch_norm`` backward with a dense input and a channels-last
strided incoming gradient (the (B, L, C) <-> (B, C, L, 1) transposes of GrapherLabel create exactly that mix).  The
product's norm layers (layers.GuardedBatchNorm2d / GuardedSyncBatchNorm) must give the closed-form gradient in both
cases; the raw torch call is checked too and only REPORTED (xfail) when it deviates, so the test documents the bug
without depending on the installed torch having it."""
import pytest
import torch
import torch.nn.functional as F


def _closed_form(x, g, gamma, eps=1e-5):
    x, g, gamma = x.double(), g.double(), gamma.double()
    dims = (0, 2, 3)
    n = x.numel() / x.shape[1]
    mean = x.mean(dims, keepdim=True)
    var = x.var(dims, unbiased=False, keepdim=True)
    xh = (x - mean) / torch.sqrt(var + eps)
    gg = g * gamma.view(1, -1, 1, 1)
    return (gg - gg.sum(dims, keepdim=True) / n - xh * (gg * xh).sum(dims, keepdim=True) / n) / torch.sqrt(var + eps)


def _case():
    torch.manual_seed(0)
    B, L, C = 4, 10, 16
    x = torch.randn(B, C, L, 1) * 2 + 1                       # dense NCHW input
    g = torch.randn(B, L, C).permute(0, 2, 1).unsqueeze(-1)    # (B, C, L, 1) view of a (B, L, C) tensor: channels-last strides
    gamma = torch.rand(C) + 0.5
    return x, g, gamma


def test_guarded_norm_layer_matches_closed_form_for_mixed_memory_formats():
    from gkgnet_amd.layers import GuardedBatchNorm2d
    x, g, gamma = _case()
    bn = GuardedBatchNorm2d(x.shape[1]).train()
    with torch.no_grad():
        bn.weight.copy_(gamma)
    xr = x.clone().requires_grad_(True)
    bn(xr).backward(g)
    want = _closed_form(x, g, gamma, bn.eps)
    assert torch.allclose(xr.grad.double(), want, atol=1e-5), float((xr.grad.double() - want).abs().max())


def test_raw_torch_batch_norm_backward_with_mixed_memory_formats():
    x, g, gamma = _case()
    xr = x.clone().requires_grad_(True)
    y = F.batch_norm(xr, None, None, gamma, torch.zeros_like(gamma), True, 0.1, 1e-5)
    y.backward(g)
    want = _closed_form(x, g, gamma)
    err = float((xr.grad.double() - want).abs().max())
    if err > 1e-4:
        pytest.xfail(f"torch {torch.__version__} CPU batch_norm backward deviates from the closed form by {err:.3g} when the "
                     f"incoming gradient is a channels-last strided view (DESIGN.md §2)")
