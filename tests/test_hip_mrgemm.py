"""SURVEY §8 row g1 (inference): gather + max(x_j - x_i) + interleave + grouped 1x1 projection + BN(eval) + GELU in one launch
(csrc/gkg_mrgemm.hip) against (i) a plain torch fp32 statement of the same definition on the same bf16-rounded operands and
(ii) the product's own three-launch form (gkg_mr_fwd_tm -> batched GEMM -> gkg_affine_act), which the reference-generated
fixtures pin.  Reference: torch_vertex.py:47-62, torch_nn.py:57-69."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (B, G, C, N, M (None = self graph), k)
CASES = [(2, 2, 80, 150, None, 9), (2, 4, 320, 324, None, 9), (3, 2, 160, 100, 25, 9), (1, 2, 400, 1296, None, 9),
         (2, 2, 640, 324, None, 9), (2, 8, 768, 70, None, 18), (2, 8, 192, 333, 90, 18), (2, 1, 48, 65, None, 5),
         (4, 4, 320, 80, 324, 9), (1, 2, 16, 64, None, 3), (1, 8, 96, 1000, 250, 18)]


def _definition(x, src, idx, G, W, a, c):
    """fp32 torch: m = max_k(src[idx] - x); u = interleave -> bf16; per conv group y = u W^T (fp32); gelu(a y + c) -> bf16"""
    B, N, C = x.shape
    cg = C // G
    s = x if src is None else src
    m = torch.empty_like(x)
    for g in range(G):
        sl = slice(g * cg, (g + 1) * cg)
        ii = idx.view(B, G, N, -1)[:, g]                                    # (B, N, k)
        nb = torch.gather(s[:, :, sl].unsqueeze(1).expand(B, N, s.shape[1], cg), 2, ii.unsqueeze(-1).expand(B, N, ii.shape[-1], cg))
        m[:, :, sl] = (nb - x[:, :, sl].unsqueeze(2)).max(2).values
    Cq = C // 4
    outs = []
    for q in range(4):
        u = torch.stack([x[:, :, q * Cq:(q + 1) * Cq], m[:, :, q * Cq:(q + 1) * Cq]], -1).reshape(B * N, 2 * Cq)
        y = u.bfloat16().float() @ W[q].bfloat16().float().t()
        outs.append(y)
    y = torch.cat(outs, 1)
    return torch.nn.functional.gelu(a * y + c).bfloat16()


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
    want = _definition(x, src, idx, G, conv.weight.detach().view(4, C // 2, C // 2), a.detach(), c.detach()).float()
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
