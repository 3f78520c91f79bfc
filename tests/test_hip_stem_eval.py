"""GPU checks of the inference stem / downsample units (fused.conv_bn_act_eval, gkg_affine_act_bf16in) against the torch modules
they replace (reference gkgnet.py:79-118 in eval mode)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _unit(cin, cout, stride, act):
    torch.manual_seed(5)
    conv = torch.nn.Conv2d(cin, cout, 3, stride=stride, padding=1).cuda()
    bn = torch.nn.BatchNorm2d(cout).cuda()
    with torch.no_grad():
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.normal_(1.0, 0.2); bn.bias.normal_()
    return conv, bn.eval(), (torch.nn.GELU() if act else None)


@pytest.mark.parametrize("act", [False, True])
@pytest.mark.parametrize("autocast", [False, True])
def test_conv_bn_act_eval_matches_modules(act, autocast):
    from gkgnet_amd import fused
    conv, bn, gelu = _unit(24, 40, 2, act)
    x = torch.randn(3, 24, 30, 26, device="cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        assert fused.conv_bn_act_eval_supported(conv, bn, gelu, x)
        ref = bn(conv(x))
        ref = gelu(ref) if act else ref
        out = fused.conv_bn_act_eval(conv, bn, gelu, x, want32=True, want16=True)
        o16 = fused.conv_bn_act_eval(conv, bn, gelu, x, want32=False, want16=True)
    assert out.dtype == torch.float32 and out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    tol = 3e-2 if autocast else 1e-4          # bf16 convolution outputs: one bf16 rounding of the pre-BN value on each side
    assert torch.allclose(out, ref.float(), atol=tol, rtol=tol), (out - ref.float()).abs().max()
    ver, x16 = out._gkg_bf16
    assert ver == out._version and x16.shape == (out.numel() // out.shape[1], out.shape[1]) and x16.dtype == torch.bfloat16
    assert torch.equal(x16.float().view(3, out.shape[2], out.shape[3], 40).permute(0, 3, 1, 2), out.to(torch.bfloat16).float())
    assert o16.dtype == torch.bfloat16 and torch.equal(o16.float(), out.to(torch.bfloat16).float())


def test_affine_act_bf16in_kernel():
    from gkgnet_amd import _lib
    from gkgnet_amd.ops import _ptr, _stream
    lib = _lib.load()
    torch.manual_seed(6)
    R, C = 777, 48
    y = torch.randn(R, C, device="cuda").to(torch.bfloat16)
    a, c = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    for act in (0, 1):
        o32 = torch.empty(R, C, device="cuda")
        o16 = torch.empty(R, C, device="cuda", dtype=torch.bfloat16)
        _lib.check(lib.gkg_affine_act_bf16in(_ptr(y), _ptr(a), _ptr(c), _ptr(o32), _ptr(o16), R, C, act, _stream()), "affine")
        ref = y.float() * a + c
        ref = torch.nn.functional.gelu(ref) if act else ref
        assert torch.allclose(o32, ref, atol=2e-6, rtol=2e-6)
        assert torch.equal(o16, o32.to(torch.bfloat16))
    with pytest.raises(_lib.GkgError):
        _lib.check(lib.gkg_affine_act_bf16in(_ptr(y), _ptr(a), _ptr(c), None, None, R, C, 0, _stream()), "affine")
