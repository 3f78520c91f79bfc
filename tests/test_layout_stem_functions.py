"""CPU checks of the small autograd Functions around the blocks (gkgnet_amd/layout.py, stem.py) against the plain torch
operators they replace (reference gkgnet.py:79-118, torch_vertex.py:194-196)."""
import pytest
import torch
import torch.nn.functional as F


@pytest.mark.gpu
def test_avg_pool_token_major_matches_avg_pool2d_forward_and_backward():
    """gkg_avgpool_tm (plain rows and the x half of an XM operand buffer) vs F.avg_pool2d, forward and backward."""
    from gkgnet_amd.layout import _AvgPoolTM
    from gkgnet_amd.fused import _xm_xview
    torch.manual_seed(0)
    dev = torch.device("cuda", 0)
    for B, H, W, C, r in ((2, 8, 8, 16, 2), (1, 12, 8, 32, 4), (2, 9, 10, 16, 4), (1, 7, 7, 48, 2)):      # the last two: floor mode
        x0 = torch.randn(B, H * W, C, device=dev)
        xr = x0.view(B, H, W, C).clone().requires_grad_(True)
        yr = F.avg_pool2d(xr.permute(0, 3, 1, 2), r, r).permute(0, 2, 3, 1)
        g = torch.randn_like(yr)
        yr.backward(g)
        for xm in (False, True):
            if xm:                                                   # x lives in the x chunks of a (B N, 2C) buffer
                XM = torch.full((B * H * W, 2 * C), float("nan"), device=dev)
                x = _xm_xview(XM, B, H * W, C)
                x.copy_(x0.view(B, H * W, 4, C // 4))
                x.requires_grad_(True)
            else:
                x = x0.clone().requires_grad_(True)
            y = _AvgPoolTM.apply(x, H, W, r)
            assert y.shape == (B, (H // r) * (W // r), C)
            assert torch.allclose(y.view_as(yr), yr, atol=1e-6), (B, H, W, C, r, xm)
            y.backward(g.reshape(y.shape))
            assert torch.allclose(x.grad.reshape(B, H, W, C), xr.grad, atol=1e-6), (B, H, W, C, r, xm)


def test_add_pos_embed_matches_broadcast_add():
    from gkgnet_amd.stem import _AddPosEmbed
    torch.manual_seed(1)
    x = torch.randn(3, 6, 5, 4, dtype=torch.float64).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    p = torch.randn(1, 6, 5, 4, dtype=torch.float64, requires_grad=True)
    xr, pr = x.detach().clone().requires_grad_(True), p.detach().clone().requires_grad_(True)
    out, ref = _AddPosEmbed.apply(x, p), xr + pr
    assert torch.allclose(out, ref)
    g = torch.randn_like(ref)
    out.backward(g)
    ref.backward(g)
    assert torch.allclose(x.grad, xr.grad) and torch.allclose(p.grad, pr.grad, atol=1e-12) and p.grad.shape == p.shape


def test_conv_before_bn_same_output_and_gradients_zero_bias_gradient():
    """The convolution in front of a train-mode BN: same output, same input / weight gradients as the module call; the bias
    gradient it does not reduce IS zero (to rounding) in the reference arrangement conv -> BN(train)."""
    from gkgnet_amd.stem import _ConvBeforeBN
    torch.manual_seed(2)
    conv = torch.nn.Conv2d(5, 8, 3, stride=2, padding=1).double()
    bn = torch.nn.BatchNorm2d(8).double().train()
    x = torch.randn(4, 5, 9, 9, dtype=torch.float64, requires_grad=True)
    xr = x.detach().clone().requires_grad_(True)
    y = bn(_ConvBeforeBN.apply(x, conv.weight, conv.bias, [2, 2], [1, 1]))
    g = torch.randn_like(y)
    gx, gw, gb = torch.autograd.grad(y, (x, conv.weight, conv.bias), g)
    yr = bn(conv(xr))
    gxr, gwr, gbr = torch.autograd.grad(yr, (xr, conv.weight, conv.bias), g)
    assert torch.allclose(y, yr, atol=1e-12)
    assert torch.allclose(gx, gxr, atol=1e-10) and torch.allclose(gw, gwr, atol=1e-10)
    assert torch.count_nonzero(gb) == 0 and gbr.abs().max() < 1e-10 * max(1.0, g.abs().sum().item())


def test_label_queries_equal_the_embedding_lookup():
    """GKGNet.forward takes the label embedding weight broadcast over the batch (label_input is arange): same values and the
    same weight gradient as nn.Embedding on the repeated index tensor."""
    torch.manual_seed(3)
    emb = torch.nn.Embedding(7, 5).double()
    idx = torch.arange(7).view(1, -1).repeat(3, 1)
    g = torch.randn(3, 7, 5, dtype=torch.float64)
    a = emb(idx)
    ga, = torch.autograd.grad(a, emb.weight, g)
    b = emb.weight.unsqueeze(0).expand(3, -1, -1)
    gb, = torch.autograd.grad(b, emb.weight, g)
    assert torch.equal(a, b) and torch.allclose(ga, gb, atol=1e-12)
