"""GKGNet backbone wiring: checkpoint-compatible state_dict on CPU; on the GPU every
block of a tiny full backbone is checked against the reference's output *for the reference's own input to that
block* (fixture F10) — a randomly initialised 16-layer k-NN network is chaotic end to end, so parity is pinned
block by block, plus a loose end-to-end check."""
import numpy as np
import pytest
import torch

from util import keyed_fill_, load_fixture


def _net(meta):
    from gkgnet_amd.backbone import GKGNet
    net = GKGNet(**meta["ctor"])
    sd = net.state_dict()
    with torch.no_grad():
        keyed_fill_(sd, seed=10)
    net.load_state_dict(sd)
    return net


def test_state_dict_matches_reference_tree():
    from gkgnet_amd.backbone import GKGNet
    meta, _ = load_fixture("f10_backbone_tiny")
    net = GKGNet(**meta["ctor"])
    sd = net.state_dict()
    ref = meta["state_shapes"]
    assert set(sd) == set(ref)
    for k, shp in ref.items():
        assert list(sd[k].shape) == shp, k


def test_registry_builds_from_config_dict():
    from gkgnet_amd.registry import build_backbone
    from gkgnet_amd.backbone import GKGNet
    net = build_backbone(dict(type="GKGNet", choice="t", k=4, k_label_gcn=4, n_classes=8, size=128))
    assert isinstance(net, GKGNet)
    assert net.layer_index == [1, 4, 11, 14]
    # per-block dilation min(idx//4+1, 49//k) and reduce ratios [4,2,1,1]  (gkgnet.py:180-183,234)
    graphers = [m[0] for m in net.backbone if isinstance(m, torch.nn.Sequential)]
    assert [g.graph_conv.d for g in graphers] == [1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3]
    assert [g.graph_conv.r for g in graphers] == [4, 4, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1]
    assert "m" in GKGNet.arch_settings and GKGNet.arch_settings["m"]["channels"] == [96, 192, 384, 768]


def _close(got, want, tol=2e-3):
    scale = max(1.0, float(np.abs(want).max()))
    err = float((got.float().cpu() - torch.from_numpy(want)).abs().max())
    return err <= tol * scale, err / scale


@pytest.mark.gpu
def test_tiny_backbone_blockwise_parity():
    meta, a = load_fixture("f10_backbone_tiny")
    net = _net(meta).cuda().eval()
    dev = lambda k: torch.from_numpy(a[k]).cuda()
    with torch.no_grad():
        x_in = net.stem(dev("img")) + net.pos_embed
        ok, err = _close(net.stem(dev("img")), a["stem"])
        assert ok, ("stem", err)
        worst = 0.0
        for bi, blk in enumerate(net.backbone):
            src = (dev("stem") + net.pos_embed) if bi == 0 else dev(f"x{bi - 1}")
            ok, err = _close(blk(src), a[f"x{bi}"])
            worst = max(worst, err)
            assert ok, (f"backbone[{bi}]", err)
        for si in range(4):
            feat = dev(f"x{net.layer_index[si]}")
            for li, gl in enumerate(net.gcn_label[si]):
                out, edge = gl(dev(f"lab_in{si}_{li}"), feat)
                ok, err = _close(out, a[f"lab_out{si}_{li}"])
                assert ok, (f"gcn_label[{si}][{li}]", err)
                assert (edge.cpu().numpy() == a[f"lab_edge{si}_{li}"]).mean() > 0.99
    assert worst < 2e-3


@pytest.mark.gpu
def test_tiny_backbone_end_to_end():
    meta, a = load_fixture("f10_backbone_tiny")
    net = _net(meta).cuda().eval()
    with torch.no_grad():
        labels, gap, edge = net(torch.from_numpy(a["img"]).cuda())
    assert labels.shape == a["label_tokens"].shape and gap.shape == a["gap"].shape
    assert edge.shape == a["edge_index"].shape and edge.dtype == torch.int64
    # End to end the numbers are NOT comparable across devices: with random weights the activations reach ~200 and
    # one fp32 near-tie flip in an early graph layer perturbs everything downstream (measured: 22 % relative
    # difference here, while every block matches to 2e-3 on the reference's own inputs and the same wiring run with
    # the oracle's operators on the CPU reproduces the reference bit for bit).  Only structure is asserted.
    assert torch.isfinite(labels).all() and torch.isfinite(gap).all()
    assert int(edge.min()) >= 0 and int(edge.max()) < 16          # last stage has 4x4 image tokens


def test_wiring_reproduces_reference_with_oracle_operators():
    """CPU: the backbone wiring driven by the oracle's k-NN / aggregation operators reproduces the reference's
    end-to-end output exactly (so any device-side difference is operator rounding, not wiring)."""
    import gkgnet_amd.graph as graph
    from oracle import torch_ref as R
    meta, a = load_fixture("f10_backbone_tiny")
    real = graph.ops
    graph.ops = type("OracleOps", (), {
        "knn_graph": staticmethod(lambda x, y, rp, k, d: R.knn_graph(x, y, rp, k, d)),
        "max_relative": staticmethod(lambda x, idx, y=None: R.max_relative(x, idx, y))})
    try:
        net = _net(meta).eval()
        with torch.no_grad():
            labels, gap, edge = net(torch.from_numpy(a["img"]))
    finally:
        graph.ops = real
    assert torch.equal(gap, torch.from_numpy(a["gap"]))
    assert torch.equal(labels, torch.from_numpy(a["label_tokens"]))
    assert np.array_equal(edge.numpy(), a["edge_index"])


def test_load_checkpoint_by_key(tmp_path):
    """f3: mmcv-style checkpoint ({'state_dict': {'backbone.*'}}) built for another input size loads by key; tensors whose
    size depends on the input resolution (pos_embed, relative_pos) are kept from the freshly built model."""
    from gkgnet_amd.backbone import GKGNet, load_checkpoint
    kw = dict(choice="t", k=4, k_label_gcn=4, n_classes=8)
    src = GKGNet(size=128, **kw)
    with torch.no_grad():
        keyed_fill_(src.state_dict(), seed=3)
    ckpt = {"state_dict": {"backbone." + k: v for k, v in src.state_dict().items()}, "meta": {}}
    ckpt["state_dict"]["head.fc1.weight"] = torch.zeros(8, 384)
    path = str(tmp_path / "ckpt.pth")
    torch.save(ckpt, path)
    dst = GKGNet(size=192, **kw)
    missing, unexpected, skipped = load_checkpoint(dst, path)
    assert not missing and not unexpected
    assert "pos_embed" in skipped and all(k == "pos_embed" or k.endswith("relative_pos") for k in skipped)
    a, b = src.state_dict(), dst.state_dict()
    for k in a:
        if k not in skipped:
            assert torch.equal(a[k], b[k]), k
