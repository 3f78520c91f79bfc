"""End-to-end on the device (SURVEY §8 f1-f4 together): an mmcv-style checkpoint is loaded by key into the GKGNet backbone
+ LabelQueryHead, a batch runs through all 16 graph layers on the HIP path, the head produces class probabilities and the
COCO metrics are computed — compared with the SAME pipeline driven by the oracle's operators on the CPU (which reproduces
the reference bit for bit, tests/test_backbone.py::test_wiring_reproduces_reference_with_oracle_operators)."""
import numpy as np
import pytest
import torch

from util import keyed_fill_

pytestmark = pytest.mark.gpu

CTOR = dict(choice="t", k=4, k_label_gcn=4, n_classes=8, size=128, drop_path=0.0)


def _build():
    from gkgnet_amd.backbone import GKGNet
    from gkgnet_amd.head import LabelQueryHead
    net = GKGNet(**CTOR)
    head = LabelQueryHead(8, 384, softmax=False, loss=dict(type="AsymmetricLoss", gamma_pos=0.0, gamma_neg=2.0, clip=0.05),
                          topk=(1, 1))
    return net, head


def _oracle_reference(imgs):
    """CPU run with the oracle's operators: calibrates the BN running statistics on the batch (momentum 1: a
    well-conditioned eval model instead of random running stats), returns the checkpoint, the eval-mode scores and the
    graphs of the eval forward in call order."""
    import gkgnet_amd.graph as graph
    from oracle import torch_ref as R
    net, head = _build()
    with torch.no_grad():
        keyed_fill_(net.state_dict(), seed=31)
        keyed_fill_(head.state_dict(), seed=32)
    graphs = []

    def knn(x, y, rp, k, d):
        e = R.knn_graph(x, y, rp, k, d)
        graphs.append(e[0].clone())
        return e
    real = graph.ops
    graph.ops = type("OracleOps", (), {"knn_graph": staticmethod(knn),
                                       "max_relative": staticmethod(lambda x, idx, y=None: R.max_relative(x, idx, y))})
    try:
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.momentum = 1.0
        net.train()
        with torch.no_grad():
            net(imgs)                                        # running statistics <- batch statistics
        for m in net.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.momentum = 0.1
        net.eval(); head.eval()
        graphs.clear()
        with torch.no_grad():
            scores = head.simple_test(net(imgs))
    finally:
        graph.ops = real
    ckpt = {"state_dict": {**{"backbone." + k: v.clone() for k, v in net.state_dict().items()},
                           **{"head." + k: v.clone() for k, v in head.state_dict().items()}}, "meta": {"epoch": 1}}
    return ckpt, scores.numpy(), [g.numpy() for g in graphs]


def test_checkpoint_forward_head_metrics_on_device(tmp_path):
    from gkgnet_amd import fused
    from gkgnet_amd.backbone import load_checkpoint
    from gkgnet_amd.coco import coco_metrics
    from gkgnet_amd.evaluation import mAP
    gen = torch.Generator().manual_seed(77)
    B = 12
    imgs = torch.randn(B, 3, 128, 128, generator=gen)
    targets = (torch.rand(B, 8, generator=gen) < 0.3).numpy().astype(np.int8)
    targets[0] = 1
    ckpt, want, graphs = _oracle_reference(imgs)
    assert len(graphs) == 16
    path = str(tmp_path / "gkgnet_tiny.pth")
    torch.save(ckpt, path)
    net, head = _build()
    missing, unexpected, skipped = load_checkpoint(net, path)          # picks the 'backbone.' keys, ignores 'head.*'
    assert not missing and not skipped and all(k.startswith("head.") for k in unexpected)
    head.load_state_dict({k[len("head."):]: v for k, v in torch.load(path)["state_dict"].items() if k.startswith("head.")})
    net.cuda().eval(); head.cuda().eval()
    real = fused.knn_graph_tm
    calls = [0]

    def forced(x, y, rp, k, d, G):
        nn_idx = torch.from_numpy(graphs[calls[0]]).cuda()
        calls[0] += 1
        center = torch.arange(nn_idx.shape[1], device="cuda").view(1, -1, 1).expand_as(nn_idx)
        return torch.stack([nn_idx, center])
    # (1) on the oracle's graphs: everything downstream of the (separately bit-exact) k-NN must agree to 1e-3
    fused.knn_graph_tm = forced
    try:
        with torch.no_grad():
            got_forced = head.simple_test(net(imgs.cuda())).cpu().numpy()
    finally:
        fused.knn_graph_tm = real
    assert calls[0] == 16, "all graph layers must run on the fused HIP path"
    assert np.abs(got_forced - want).max() < 1e-3
    assert abs(mAP(got_forced, targets) - mAP(want, targets)) < 1e-6
    m1, m2 = coco_metrics(targets, got_forced), coco_metrics(targets, want)
    assert all(abs(m1[k] - m2[k]) < 1e-9 or (np.isnan(m1[k]) and np.isnan(m2[k])) for k in m2)
    # (2) free-running (the product's own graphs): near-tie flips may move single scores; the metrics stay close
    with torch.no_grad():
        got = head.simple_test(net(imgs.cuda())).cpu().numpy()
    assert np.isfinite(got).all() and got.shape == want.shape
    assert (np.abs(got - want) < 0.05).mean() >= 0.9
    assert abs(mAP(got, targets) - mAP(want, targets)) < 10.0


def test_bf16_inference_shortcuts_keep_the_forward():
    """bf16 autocast inference of the whole backbone with the round-2 layout / epilogue shortcuts on (channels-last
    chaining, bf16 copy emitted by a block's last kernel, BN folded into the FFN GEMM's bias + GELU epilogue) against the
    same forward with them off, on the SAME graphs (the second run replays the first run's neighbour lists: with random
    weights a near-tie flip would otherwise dominate the comparison): equal within what bf16 rounding allows."""
    from gkgnet_amd import fused
    torch.manual_seed(5)
    net, _ = _build()
    with torch.no_grad():
        keyed_fill_(net.state_dict(), seed=41)
    net = net.cuda()
    imgs = torch.randn(4, 3, 128, 128, device="cuda")
    net.train()
    with torch.no_grad():
        for _ in range(2):
            net(imgs)                                     # a well-conditioned eval model: calibrate the running statistics
    net.eval()
    outs, graphs = {}, []
    real = fused.knn_graph_tm
    saved = (fused.CHANNELS_LAST, fused.FOLD_EPILOGUE)
    try:
        for mode in ("off", "on"):
            fused.CHANNELS_LAST = fused.FOLD_EPILOGUE = (mode == "on")
            it = iter(graphs)

            def knn(*a, **k):
                if mode == "off":
                    e = real(*a, **k)
                    graphs.append(e)
                    return e
                return next(it)
            fused.knn_graph_tm = knn
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                labels, gap, edge = net(imgs)
            outs[mode] = (labels.float(), gap.float())
    finally:
        fused.knn_graph_tm = real
        fused.CHANNELS_LAST, fused.FOLD_EPILOGUE = saved
    assert len(graphs) == 16
    for a, b in zip(outs["on"], outs["off"]):
        assert a.shape == b.shape and torch.isfinite(a).all() and torch.isfinite(b).all()
        scale = b.abs().mean().item()
        assert (a - b).abs().mean().item() <= 0.02 * scale + 1e-3, ((a - b).abs().mean().item(), scale)


def test_inference_caches_follow_the_running_statistics():
    """eval -> train steps (the fused kernels update the running statistics through raw pointers) -> eval: the cached
    eval-mode BN coefficients and folded weights must be rebuilt."""
    from gkgnet_amd.backbone import FFN
    torch.manual_seed(3)
    ffn = FFN(32, 64, act="gelu").cuda()
    x = torch.randn(2, 32, 10, 10, device="cuda").contiguous(memory_format=torch.channels_last)

    def infer():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            return ffn(x).float()

    ffn.eval()
    y0 = infer()
    ffn.train()
    for _ in range(3):
        ffn(torch.randn(2, 32, 10, 10, device="cuda") * 3 + 1)     # moves running_mean / running_var a lot
    ffn.eval()
    y1 = infer()
    from gkgnet_amd import fused
    fused.ENABLED = False
    try:
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            want = ffn(x).float()
    finally:
        fused.ENABLED = True
    assert (y1 - y0).abs().mean() > 0.05 * y0.abs().mean()       # the statistics did change the function
    assert (y1 - want).abs().mean() <= 0.02 * want.abs().mean() + 1e-3
