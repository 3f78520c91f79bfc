import sys, os, torch
sys.path.insert(0, os.getcwd())
from gkgnet_amd import fused, relpos
torch.manual_seed(0)
def t(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1000
for name, (B, G, C, N, M, r, n) in {"s1": (32, 2, 80, 20736, 1296, 4, 144 * 144), "s2": (32, 2, 160, 5184, 1296, 2, 72 * 72)}.items():
    x = torch.randn(B, N, C, device="cuda"); y = torch.randn(B, M, C, device="cuda")
    rp = relpos.build_relative_pos(C, n, r).cuda()
    rp_rand = -torch.rand(1, N, M, device="cuda")
    print(name, "relpos", tuple(rp.shape), float(rp.min()), float(rp.max()))
    print(name, "with model relpos : %.1f us" % t(lambda: fused.knn_graph_tm16(x, y, rp, 9, 1, G)))
    print(name, "with random relpos: %.1f us" % t(lambda: fused.knn_graph_tm16(x, y, rp_rand, 9, 1, G)))
    print(name, "without relpos    : %.1f us" % t(lambda: fused.knn_graph_tm16(x, y, None, 9, 1, G)))
