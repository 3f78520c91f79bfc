"""Fused vs composable Grapher block (forced graph) at one shape: fraction of input-gradient elements within 2e-3 (+ 2e-3 rel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gkgnet_amd import fused
from gkgnet_amd.grapher import Grapher
import gkgnet_amd.graph as graph

C, d, H, B, G = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
torch.manual_seed(0)
g = Grapher(C, 9, d, "mr", "gelu", "batch", True, False, 0.2, 1, n=H * H, relative_pos=True, use_multi_group=True, num_group=G).cuda().train()
x = torch.randn(B, C, H, H, device="cuda")
cot = torch.randn_like(x)
rec = {}
real_tm = fused.knn_graph_tm
def recording(*a, **k):
    rec["edge"] = real_tm(*a, **k)
    return rec["edge"]
real_ops = graph.ops
forced = type("F", (), {"knn_graph": staticmethod(lambda *a, **k: rec["edge"]), "max_relative": staticmethod(real_ops.max_relative)})
res = []
for enabled in (True, False):
    fused.ENABLED = enabled
    fused.knn_graph_tm = recording
    graph.ops = real_ops if enabled else forced
    xg = x.clone().requires_grad_(True)
    out = g(xg)
    out.backward(cot)
    res.append((out.detach(), xg.grad.clone(), {n: p.grad.clone() for n, p in g.named_parameters() if p.grad is not None}))
    g.zero_grad(set_to_none=True)
    fused.ENABLED = True; fused.knn_graph_tm = real_tm; graph.ops = real_ops
(o1, d1, p1), (o2, d2, p2) = res
bad = (d1 - d2).abs() > 2e-3 + 2e-3 * d2.abs()
print(os.environ.get("GKG_GEMM_MATH"), os.environ.get("GKG_DISABLE"), "out maxdiff", float((o1 - o2).abs().max()), "okg", 1 - bad.float().mean().item(),
      "rel norm", ((d1 - d2).norm() / d2.norm()).item(), "max", float((d1 - d2).abs().max()), "dmax", float(d2.abs().max()))
idx = bad.nonzero()
if len(idx):
    print(" bad per image", [int((idx[:, 0] == b).sum()) for b in range(B)], "distinct channels", len(idx[:, 1].unique()), "distinct pixels", len((idx[:, 2] * H + idx[:, 3]).unique()))
for n in p1:
    e = float((p1[n] - p2[n]).abs().max() / (p2[n].abs().max() + 1e-12))
    if e > 1e-3:
        print("  param", n, e)
