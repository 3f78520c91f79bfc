#!/usr/bin/env python
"""Micro-benchmark of the HIP operators alone (library-side HIP-event timing per kernel).
    python tools/bench_ops.py [shape-name ...]
"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gkgnet_amd import _lib, ops
from gkgnet_amd.relpos import build_relative_pos

SHAPES = {   # name: (BG, c, N, M(None=self), k, d, relpos_C)
    "cfg2_grapher": (128, 80, 324, None, 9, 1, 320),
    "cfg2_grapher_norp": (128, 80, 324, None, 9, 1, None),
    "cfg2_bg42": (42, 80, 324, None, 9, 1, 320),
    "cfg2_bg84": (84, 80, 324, None, 9, 1, 320),
    "cfg2_bg256": (256, 80, 324, None, 9, 1, 320),
    "cfg2_label": (128, 80, 80, 324, 9, 1, None),
    "cfg2ref_grapher": (64, 320, 324, None, 9, 3, 640),
    "stage3_d2": (64, 200, 1296, None, 9, 2, 400),
    "stage2": (64, 80, 5184, 1296, 9, 1, "rand"),
    "stage1": (64, 40, 20736, 1296, 9, 1, "rand"),
    "stage1_norp": (64, 40, 20736, 1296, 9, 1, None),
    "label_stage1": (64, 40, 80, 20736, 9, 1, None),
    "stage3_d3": (64, 200, 1296, None, 9, 3, 400),
    # pvig_m @ 768, B = 16, G = 8 (BASELINE config 5)
    "m1": (128, 12, 36864, 2304, 18, 1, "rand"),
    "m2": (128, 24, 9216, 2304, 18, 1, "rand"),
    "m3": (128, 48, 2304, None, 18, 2, "rand"),
    "m4": (128, 96, 576, None, 18, 2, "rand"),
    "m1_label": (128, 12, 80, 36864, 18, 1, None),
}

def run(name, iters=20):
    BG, c, N, M, k, d, rpc = SHAPES[name]
    torch.manual_seed(0)
    x = torch.randn(BG, c, N, device="cuda")
    y = None if M is None else torch.randn(BG, c, M, device="cuda")
    Mk = N if M is None else M
    rp = None
    if rpc == "rand":
        rp = -torch.rand(1, N, Mk, device="cuda")
    elif rpc is not None:
        rp = build_relative_pos(rpc, N, 1).cuda()
    for _ in range(3):
        e = ops.knn_graph(x, y, rp, k, d)
    xg = x.clone().requires_grad_(True)
    yg = None if y is None else y.clone().requires_grad_(True)
    g = torch.randn(BG, c, N, device="cuda")
    for _ in range(2):
        ops.max_relative(xg, e[0], yg).backward(g)
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(iters):
        e = ops.knn_graph(x, y, rp, k, d)
        ops.max_relative(xg, e[0], yg).backward(g)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    pr = _lib.prof_read()
    flops = 2.0 * BG * c * N * Mk
    es = 4
    b_mr = es * BG * c * N * 2 + (es * BG * c * Mk if M is not None else 0) + 8 * BG * N * k + BG * c * N
    b_bw = es * BG * c * N * 2 + 8 * BG * N * k + BG * c * N + (es * BG * c * Mk if M is not None else 0)
    out = {"shape": name}
    for kn, (ms, cnt) in pr.items():
        if cnt:
            out[kn + "_us"] = round(1e3 * ms / cnt * (cnt / iters), 1)
    if "knn_tile_us" in out:
        out["knn_TF"] = round(flops / out["knn_tile_us"] / 1e6, 1)
        out["knn_frac"] = round(out["knn_TF"] / 157.3, 3)
    out["mr_fwd_GBs"] = round(b_mr / out["mr_fwd_us"] / 1e3, 0)
    out["mr_bwd_GBs"] = round(b_bw / out["mr_bwd_us"] / 1e3, 0)
    print(json.dumps(out), flush=True)

if __name__ == "__main__":
    for n in (sys.argv[1:] or list(SHAPES)):
        run(n)
