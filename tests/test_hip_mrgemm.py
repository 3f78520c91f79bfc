"""SURVEY §8 row g1 (inference): gather + max(x_j - x_i) + interleave + grouped 1x1 projection + BN(eval) + GELU in one launch
(csrc/gkg_mrgemm.hip) against (i) the C ORACLE's aggregation (oracle/c_oracle.mr_fwd) followed by a dense grouped 1x1 convolution
(F.conv2d, groups=4) of the oracle's interleaved output on the same bf16-rounded operands and
(ii) the product's own three-launch form (gkg_mr_fwd_tm -> batched GEMM -> gkg_affine_act), which the reference-generated
fixtures pin.  Reference: torch_vertex.py:47-62, torch_nn.py:57-69."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (B, G, C, N, M (None = self graph), k)
CASES = [(2, 2, 80, 150, None, 9), (2, 4, 320, 324, None, 9), (3, 2, 160, 100, 25, 9), (1, 2, 400, 1296, None, 9),
         (2, 2, 640, 324, None, 9), (2, 8, 768, 70, None, 18), (2, 8, 192, 333, 90, 18), (2, 1, 48, 65, None, 5),
         (4, 4, 320, 80, 324, 9), (1, 2, 16, 64, None, 3), (1, 8, 96, 1000, 250, 18),
         (2, 3, 288, 100, None, 9)]       # G = 3: a conv group (C/4 channels) straddles two k-NN groups (C/3) — ADVICE r3


def _definition(x, src, idx, G, conv_weight, a, c):
    """Expected value from the ORACLE: m = oracle mr_fwd (C, bit-exact contract of max_k(src[idx] - x), torch_vertex.py:49-54);
    [x, m] interleaved exactly like the reference's cat (torch_vertex.py:61), rounded to bf16 (the kernel's operand type);
    the reference's BasicConv as a dense F.conv2d(groups=4) with the bf16-rounded weight in fp32 (torch_nn.py:57-69);
    gelu(a y + c) -> bf16."""
    from oracle import c_oracle as O
    B, N, C = x.shape
    cg = C // G
    xc = x.permute(0, 2, 1).reshape(B, G, cg, N).reshape(B * G, cg, N).cpu().numpy()          # the reference's (B*G, c, N)
    sc = None if src is None else src.permute(0, 2, 1).reshape(B * G, cg, src.shape[1]).cpu().numpy()
    m, _ = O.mr_fwd(xc, sc, idx.cpu().numpy())
    m = torch.from_numpy(m).reshape(B, C, N, 1)
    xr = x.permute(0, 2, 1).reshape(B, C, N, 1).cpu()
    u = torch.cat([xr.unsqueeze(2), m.unsqueeze(2)], dim=2).reshape(B, 2 * C, N, 1)               # torch_vertex.py:61
    y = torch.nn.functional.conv2d(u.bfloat16().float(), conv_weight.cpu().bfloat16().float(), None, groups=4)
    y = y.reshape(B, 2 * C, N).permute(0, 2, 1).reshape(B * N, 2 * C)
    return torch.nn.functional.gelu(a.cpu() * y + c.cpu()).bfloat16().cuda()


@pytest.mark.parametrize("B,G,C,N,M,k", CASES)
def test_fused_aggregation_projection_matches_its_definition(B, G, C, N, M, k):
    from gkgnet_amd import fused
    gen = torch.Generator(device="cuda").manual_seed(C + N)
    x = torch.randn(B, N, C, device="cuda", generator=gen)
    src = None if M is None else torch.randn(B, M, C, device="cuda", generator=gen)
    Mk = N if M is None else M
    idx = torch.randint(0, Mk, (B * G, N, k), device="cuda", generator=gen)
    conv = torch.nn.Conv2d(2 * C, 2 * C, 1, groups=4).cuda()
    bn = torch.nn.BatchNorm2d(2 * C).cuda().eval()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.2); bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    got = fused.mr_grouped_linear_eval(x, src, idx, G, conv, bn).float()
    inv = torch.rsqrt(bn.running_var + bn.eps)
    a = bn.weight * inv
    c = bn.bias + a * (conv.bias - bn.running_mean)
    want = _definition(x, src, idx, G, conv.weight.detach(), a.detach(), c.detach()).float()
    assert got.shape == want.shape == (B * N, 2 * C)
    # bf16 outputs of fp32 accumulations that differ in summation order: a rounding boundary may flip -> 1 bf16 ulp
    err = (got - want).abs()
    tol = 2.0 ** -7 * want.abs() + 1e-3
    assert (err <= tol).float().mean().item() == 1.0, (err.max().item(), (err > tol).sum().item())
    assert (got == want).float().mean().item() > 0.97


def test_block_forward_uses_the_fused_launch_and_matches_the_three_launch_form(monkeypatch):
    """Grapher + GrapherLabel, eval, bf16 autocast: the block with the fused launch equals the block with
    gkg_mr_fwd_tm -> GEMM -> gkg_affine_act to bf16 resolution, and the fused launch is the one that runs."""
    from gkgnet_amd import fused, layers
    from gkgnet_amd.grapher import Grapher, GrapherLabel
    from tests.util import keyed_fill_
    layers.norm_cfg["type"] = "BN"
    torch.manual_seed(0)
    C, G, H, L = 160, 2, 16, 80
    g = Grapher(C, 9, 2, "mr", "gelu", "batch", True, False, 0.2, 2, n=H * H, drop_path=0.0, relative_pos=True,
                use_multi_group=True, num_group=G)
    gl = GrapherLabel(C, 9, 1, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, drop_path=0.0, relative_pos=False,
                      num_nodes=L, use_multi_group=True, num_group=G)
    for m in (g, gl):
        sd = m.state_dict()
        keyed_fill_(sd)
        m.load_state_dict(sd)
        m.cuda().eval()
    x = torch.randn(2, C, H, H, device="cuda")
    e = torch.randn(2, L, C, device="cuda")
    calls = [0]
    real = fused.mr_grouped_linear_eval

    def counting(*a, **k):
        calls[0] += 1
        return real(*a, **k)
    monkeypatch.setattr(fused, "mr_grouped_linear_eval", counting)
    outs = []
    for on in (True, False):
        monkeypatch.setattr(fused, "MR_GEMM", on)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            o = g(x)
            e2, idx = gl(e, o)
        outs.append((o.float(), e2.float(), idx))
    assert calls[0] == 2
    (o1, e1, i1), (o2, e2_, i2) = outs
    assert torch.equal(i1, i2)
    assert torch.allclose(o1, o2, atol=3e-2, rtol=3e-2) and (o1 - o2).abs().mean() < 2e-3
    assert torch.allclose(e1, e2_, atol=3e-2, rtol=3e-2) and (e1 - e2_).abs().mean() < 2e-3
