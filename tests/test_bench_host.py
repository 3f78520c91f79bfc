"""Host-side pieces of bench.py that run without a GPU: the whole-backbone CPU baseline (a CPU copy of the network with the
graph operators routed to the oracle) and the workload tables."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_backbone_cpu_baseline_runs_forward_and_train_step_on_the_oracle():
    import bench
    from gkgnet_amd import layers
    from gkgnet_amd.backbone import GKGNet
    from gkgnet_amd.head import LabelQueryHead
    import gkgnet_amd.graph as graph
    layers.norm_cfg["type"] = "BN"
    torch.manual_seed(0)
    net = GKGNet(choice="t", k=4, k_label_gcn=4, n_classes=8, size=128)
    head = LabelQueryHead(8, GKGNet.arch_settings["t"]["channels"][-1])
    img = torch.randn(2, 3, 128, 128)
    tgt = (torch.rand(2, 8) < 0.3).float()
    real = graph.ops
    fwd = bench.cpu_baseline_backbone(net.eval(), None, img[:1], None, "forward", budget_s=0.5, threads=2)
    assert fwd["value"] > 0 and fwd["kind"] == "port" and fwd["cores"] == 2 and "oracle/torch_ref.py" in fwd["sample"]
    trn = bench.cpu_baseline_backbone(net.train(), head.train(), img, tgt, "train", budget_s=0.5, threads=2)
    assert trn["value"] > 0 and "backward" in trn["sample"]
    assert graph.ops is real                     # the operator swap is undone
    assert all(p.grad is None for p in net.parameters())      # the timed copy, not the caller's network, was stepped


def test_workload_tables_cover_every_baseline_config():
    import bench
    assert {"cfg2", "cfg2ref", "stage3", "stage1"} <= set(bench.WORKLOADS)
    assert set(bench.BACKBONE_WORKLOADS) == {"cfg3", "cfg4", "cfg5"}
    assert bench.BACKBONE_WORKLOADS["cfg5"]["kw"]["num_group"] == 8 and bench.BACKBONE_WORKLOADS["cfg5"]["B"] == 16
    assert bench.WORKLOADS["stage1"]["H"] ** 2 == 20736 and bench.WORKLOADS["stage1"]["r"] == 4
