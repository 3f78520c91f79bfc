"""BASELINE configs 3 and 5 at their REAL layer shapes on the device (reduced batch): the full GKGNet-576 (pvig_s) and the
pvig_m @ 768 / k = 18 / G = 8 backbones are built, every one of their 16 / 26 graph layers must take the fused HIP path,
and one block per distinct shape is compared — fused token-major path vs the composable per-op path (which is pinned to
the reference's fixtures op by op) on identical inputs and weights, fp32, train-mode batch statistics, forward + input
gradient within the 1e-3 contract.  Finishes with the configs' own run mode: eval under bf16 autocast, finite outputs and
in-range graphs."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(choice, size, k, G, n_classes=80):
    os.environ["GKG_RELPOS_DEVICE"] = "cuda"                  # relative_pos constants built on the device (seconds, not 40 s)
    from gkgnet_amd import layers
    from gkgnet_amd.backbone import GKGNet
    layers.norm_cfg["type"] = "BN"
    torch.manual_seed(0)
    return GKGNet(choice=choice, k=k, k_label_gcn=k, n_classes=n_classes, size=size, num_group=G).cuda()


def _block_pairs(net):
    """(name, Grapher block, input shape) for the first Grapher of every (channels, dilation, r) combination + the label
    graph convs."""
    from gkgnet_amd.grapher import Grapher
    seen, out = set(), []
    for i, blk in enumerate(net.backbone):
        if isinstance(blk, torch.nn.Sequential) and isinstance(blk[0], Grapher):
            g = blk[0]
            key = (g.channels, g.graph_conv.d, g.graph_conv.r)
            if key not in seen:
                seen.add(key)
                out.append((f"backbone[{i}] C={key[0]} d={key[1]} r={key[2]}", g))
    return out


@pytest.mark.parametrize("cfg", ["cfg3_pvig_s_576", "cfg5_pvig_m_768"])
def test_real_shape_blocks_fused_vs_composable_and_autocast_forward(cfg):
    from gkgnet_amd import fused
    if cfg.startswith("cfg3"):
        net, size, B = _build("s", 576, 9, 2), 576, 2
    else:
        net, size, B = _build("m", 768, 18, 8), 768, 1
    n_graph = sum(1 for m in net.modules() if type(m).__name__ == "DenseDilatedKnnGraph")
    assert n_graph == (16 if cfg.startswith("cfg3") else 26)
    # ---- (1) one block per distinct shape: fused vs composable, fp32 train mode
    side = size // 4
    shapes = {}
    hw = side
    ch = net.arch_settings["s" if cfg.startswith("cfg3") else "m"]["channels"]
    for stage, c in enumerate(ch):
        shapes[c] = hw
        hw //= 2
    torch.manual_seed(1234)
    for name, g in _block_pairs(net):
        g.train()
        C = g.channels
        x = torch.randn(B, C, shapes[C], shapes[C], device="cuda")
        cot = torch.randn_like(x)
        res = []
        rec = {}
        real_tm = fused.knn_graph_tm

        def recording(*a, **k):
            rec["edge"] = real_tm(*a, **k)
            return rec["edge"]
        import gkgnet_amd.graph as graph
        real_ops = graph.ops
        # the composable run is given the fused run's graph: the k-NN operator is pinned bit-exactly at these very shapes
        # (tests/test_hip_config_shapes.py); fc1's two implementations differ by ~1e-6, enough to flip fp32 near-ties
        # among 36 864 x 18 neighbour slots, which would otherwise blur an element-wise comparison
        forced_ops = type("ForcedGraph", (), {"knn_graph": staticmethod(lambda *a, **k: rec["edge"]),
                                              "max_relative": staticmethod(real_ops.max_relative)})
        for enabled in (True, False):
            fused.ENABLED = enabled
            fused.knn_graph_tm = recording
            graph.ops = real_ops if enabled else forced_ops
            try:
                xg = x.clone().requires_grad_(True)
                out = g(xg)
                out.backward(cot)
                res.append((out.detach(), xg.grad.clone()))
            finally:
                fused.ENABLED = True
                fused.knn_graph_tm = real_tm
                graph.ops = real_ops
            g.zero_grad(set_to_none=True)
        (o1, d1), (o2, d2) = res
        assert torch.allclose(o1, o2, atol=1e-3, rtol=1e-3), (name, float((o1 - o2).abs().max()))
        # a near-tie inside max_k(x_j - x_i) (two neighbours equal in one channel to ~1e-6: the two paths' fc1 outputs differ
        # by that much) routes ONE gradient element to the other neighbour — which fc1's input gradient then spreads over all C
        # channels of the (up to three) tokens involved.  So the comparison is per TOKEN: all but a handful of tokens agree in
        # every channel (expected flips ~1e-6 per (token, channel) maximum), and the difference is bounded in norm.  (Round 5:
        # the former element-wise ">= 99.9 %" bar counted one flip at B = 2 as 0.3 % of the elements.)
        bad_tok = ((d1 - d2).abs() > 2e-3 + 2e-3 * d2.abs()).any(dim=1)                 # (B, H, W)
        flips_allowed = 2 + int(3e-6 * d1.numel())
        assert int(bad_tok.sum()) <= 3 * flips_allowed, (name, int(bad_tok.sum()), flips_allowed)
        assert ((d1 - d2).norm() / d2.norm()).item() < 2e-2, name
    # ---- (2) the config's own mode: eval, bf16 autocast, every graph layer on the fused HIP path — and the graphs it builds
    # there are the CONTRACT's, index for index: the first k-NN call of every distinct (queries, keys, channels, list) shape
    # is re-evaluated by the C oracle on the very tokens the kernel saw ("bit-exact neighbor indices" in
    # the run mode of cfg3 / cfg5; the bf16 contraction is opt-in, GKG_ENABLE=knn_bf16)
    import numpy as np
    from oracle import c_oracle as O
    assert fused.KNN_BF16 is False, "the index-exact k-NN must be the default under autocast"
    net.eval()
    calls = {"n": 0}
    seen = {}
    real = fused.knn_graph_tm

    def counting(x, y, rp, k, d, G):
        calls["n"] += 1
        edge = real(x, y, rp, k, d, G)
        key = (tuple(x.shape), None if y is None else tuple(y.shape), k, d, G, rp is not None)
        if key not in seen:
            seen[key] = (x.detach().clone(), None if y is None else y.detach().clone(),
                         None if rp is None else rp.detach().clone(), k, d, G, edge[0].clone())
        return edge
    fused.knn_graph_tm = counting
    try:
        img = torch.randn(B, 3, size, size, device="cuda")
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            labels, gap, edge = net(img)
    finally:
        fused.knn_graph_tm = real
    assert calls["n"] == n_graph, "all graph layers must run on the fused HIP path"
    assert labels.shape == (B, 80, ch[-1]) and gap.shape == (B, ch[-1])
    assert torch.isfinite(labels.float()).all() and torch.isfinite(gap.float()).all()
    last_tokens = (size // 32) ** 2
    assert edge.dtype == torch.int64 and int(edge.min()) >= 0 and int(edge.max()) < last_tokens
    assert len(seen) >= 8, sorted(seen)               # 4 stages x (Grapher, GrapherLabel) at least

    def groups_major(t, G):                           # (B, T, C) token-major -> the reference's (B*G, c, T)
        Bt, T, C = t.shape
        return t.float().view(Bt, T, G, C // G).permute(0, 2, 3, 1).reshape(Bt * G, C // G, T).cpu().numpy()
    for key, (x, y, rp, k, d, G, nn_idx) in seen.items():
        assert x.dtype == torch.float32, key          # the k-NN reads fp32 tokens under autocast (fc1's fp32 output)
        want, _ = O.knn(groups_major(x, G), None if y is None else groups_major(y, G),
                        None if rp is None else rp.float().reshape(-1, rp.shape[-1]).cpu().numpy(), k, d)
        got = nn_idx.cpu().numpy()
        assert np.array_equal(got, want), (key, float((got != want).mean()))
