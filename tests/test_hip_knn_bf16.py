"""k-NN graph with the distance contraction on the bf16 matrix cores (GKG_KNN_BF16_CONTRACT), against an fp64 evaluation of
the same definition: d = |x^|^2 + |y^|^2 - 2 bf16(x^) . bf16(y^) (+ relative_pos), x^ the L2-normalised token.  Not part of
the bit-exact index contract: a neighbour may differ from the fp64 ranking only where the two distances are within 1e-3
(bf16 roundings of a few normalised features flipping: the test normalises in a different summation order), and almost all must agree exactly."""
import numpy as np
import pytest
import torch

from tests.util import check_indices

pytestmark = pytest.mark.gpu

# (B, G, c, N, M (None = self graph), k, d, relpos)
CASES = [(2, 4, 80, 324, None, 9, 1, True), (2, 4, 80, 80, 324, 9, 1, False), (1, 2, 200, 1296, None, 9, 2, True),
         (1, 2, 320, 324, None, 9, 3, True), (1, 2, 40, 5184, 1296, 9, 1, True), (1, 8, 48, 2304, None, 18, 2, True),
         (1, 1, 16, 100, 333, 5, 1, False), (2, 2, 24, 77, None, 3, 2, True)]


def _reference(x, y, rp, k, d, G):
    B, N, C = x.shape
    c = C // G
    xg = x.view(B, N, G, c).permute(0, 2, 1, 3).reshape(B * G, N, c)
    yg = xg if y is None else y.view(B, -1, G, c).permute(0, 2, 1, 3).reshape(B * G, -1, c)
    xn = torch.nn.functional.normalize(xg, dim=-1)
    yn = torch.nn.functional.normalize(yg, dim=-1)
    inner = xn.bfloat16().double() @ yn.bfloat16().double().transpose(1, 2)
    dist = (xn.double() ** 2).sum(-1, keepdim=True) - 2 * inner + (yn.double() ** 2).sum(-1).unsqueeze(1)
    if rp is not None:
        dist = dist + rp.double()
    topd, topi = torch.topk(dist, k * d + 1 if k * d + 1 <= dist.shape[-1] else k * d, dim=-1, largest=False)
    return topd.cpu().numpy(), topi.cpu().numpy()


@pytest.mark.parametrize("select", ["auto", "buffered"])
@pytest.mark.parametrize("B,G,c,N,M,k,d,relpos", CASES)
def test_bf16_contraction_matches_its_definition(B, G, c, N, M, k, d, relpos, select, monkeypatch):
    from gkgnet_amd import fused
    if select == "buffered":
        monkeypatch.setenv("GKG_KNN_SELECT", "buffered")
    gen = torch.Generator(device="cuda").manual_seed(c * 1000 + N)
    x = torch.randn(B, N, G * c, device="cuda", generator=gen)
    y = None if M is None else torch.randn(B, M, G * c, device="cuda", generator=gen)
    Mk = N if M is None else M
    rp = -torch.rand(1, N, Mk, device="cuda", generator=gen) if relpos else None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        monkeypatch.setattr(fused, "KNN_BF16", True)                  # opt-in mode (GKG_ENABLE=knn_bf16)
        edge = fused.knn_graph_tm(x, y, rp, k, d, G)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        monkeypatch.setattr(fused, "KNN_BF16", False)
        edge32 = fused.knn_graph_tm(x, y, rp, k, d, G)                # the fp32 contraction on the same inputs
    topd, topi = _reference(x, y, rp, k, d, G)
    got = edge[0].cpu().numpy()
    want = topi[:, :, 0:k * d:d]
    swaps = check_indices(got, want, topd, topi, dilation=d, tol=1e-3)
    assert swaps <= 0.01 * got.size, (swaps, got.size)
    assert torch.equal(edge[1], edge32[1])
    # and it is a bf16-level perturbation of the fp32 graph: with random features the candidates are nearly equidistant, so
    # sets move, but the neighbours it picks are as close (in exact fp64 distance of the unrounded tokens) as fp32's
    B_, N_, C_ = x.shape
    xg = torch.nn.functional.normalize(x.view(B_, N_, G, c).permute(0, 2, 1, 3).reshape(B_ * G, N_, c).double(), dim=-1)
    yg = xg if y is None else torch.nn.functional.normalize(
        y.view(B_, -1, G, c).permute(0, 2, 1, 3).reshape(B_ * G, -1, c).double(), dim=-1)
    dist = 2 - 2 * xg @ yg.transpose(1, 2)
    if rp is not None:
        dist = dist + rp.double()
    d16 = torch.gather(dist, 2, edge[0]).mean().item()
    d32 = torch.gather(dist, 2, edge32[0]).mean().item()
    assert d16 <= d32 + 2e-3, (d16, d32)
