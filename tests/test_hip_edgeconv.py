"""EdgeConv2d through the HIP aggregation (csrc/gkg_edge.hip) against the literal form of the reference
(torch_vertex.py:82-101: gather x_j, cat[x_i, x_j - x_i], grouped 1x1 conv + norm + act on (B, 2C, N, k), max over k)
evaluated with torch ops on the same device: outputs, input / source / parameter gradients and BN running statistics."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _literal(mod, x, edge_index, y=None):
    bg, c = x.shape[:2]
    xt = x.reshape(bg, c, -1)
    src = xt if y is None else y.reshape(bg, c, -1)
    idx = edge_index[0]
    n, k = idx.shape[1:]
    x_j = torch.gather(src, 2, idx.reshape(bg, 1, n * k).expand(bg, c, n * k)).reshape(bg, c, n, k)
    x_i = xt.unsqueeze(-1).expand(-1, -1, -1, k)
    return mod.nn(torch.cat([x_i, x_j - x_i], dim=1)).max(dim=-1, keepdim=True).values


@pytest.mark.parametrize("bipartite", [False, True])
@pytest.mark.parametrize("norm,act,train", [("batch", "gelu", True), ("batch", "relu", True), ("batch", "gelu", False),
                                            (None, "relu", True), (None, "gelu", True)])
def test_edgeconv_hip_matches_literal_form(norm, act, train, bipartite):
    from gkgnet_amd import layers
    from gkgnet_amd.graph import EdgeConv2d
    old = dict(layers.norm_cfg)
    layers.norm_cfg.update(type="BN")
    try:
        torch.manual_seed(7)
        B, C, N, M, k, out = 3, 24, 50, 37, 6, 40
        mod = EdgeConv2d(C, out, act, norm, True).cuda()
        with torch.no_grad():
            for p in mod.parameters():
                p.add_(0.1 * torch.randn_like(p))
            if norm:
                mod.nn[1].running_mean.normal_(0, 0.2); mod.nn[1].running_var.uniform_(0.5, 1.5)
        ref = copy.deepcopy(mod)
        mod.train(train); ref.train(train)
        assert mod._hip_plan(torch.zeros(1, C, 1, 1, device="cuda")) is not None
        x = torch.randn(B, C, N, 1, device="cuda", requires_grad=True)
        y = torch.randn(B, C, M, 1, device="cuda", requires_grad=True) if bipartite else None
        Mk = M if bipartite else N
        idx = torch.stack([torch.randperm(Mk, device="cuda")[:k] for _ in range(B * N)]).view(B, N, k)
        edge = torch.stack([idx, torch.arange(N, device="cuda").view(1, N, 1).expand(B, N, k)])
        outp = mod(x, edge, y)
        x2 = x.detach().clone().requires_grad_(True)
        y2 = None if y is None else y.detach().clone().requires_grad_(True)
        want = _literal(ref, x2, edge, y2)
        assert outp.shape == want.shape
        assert torch.allclose(outp, want, atol=2e-5, rtol=1e-5), float((outp - want).abs().max())
        g = torch.randn_like(want)
        outp.backward(g); want.backward(g)
        assert torch.allclose(x.grad, x2.grad, atol=5e-5, rtol=1e-4), float((x.grad - x2.grad).abs().max())
        if bipartite:
            assert torch.allclose(y.grad, y2.grad, atol=5e-5, rtol=1e-4), float((y.grad - y2.grad).abs().max())
        for (name, p), (_, q) in zip(mod.named_parameters(), ref.named_parameters()):
            gp = torch.zeros_like(p) if p.grad is None else p.grad
            gq = torch.zeros_like(q) if q.grad is None else q.grad
            assert torch.allclose(gp, gq, atol=2e-4, rtol=1e-4), (name, float((gp - gq).abs().max()))
        if norm:
            assert torch.allclose(mod.nn[1].running_mean, ref.nn[1].running_mean, atol=1e-5)
            assert torch.allclose(mod.nn[1].running_var, ref.nn[1].running_var, atol=1e-5, rtol=1e-5)
            assert int(mod.nn[1].num_batches_tracked) == int(ref.nn[1].num_batches_tracked)
    finally:
        layers.norm_cfg.clear(); layers.norm_cfg.update(old)
