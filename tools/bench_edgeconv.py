#!/usr/bin/env python
"""EdgeConv2d forward + backward: the HIP aggregation path (per-node projections + csrc/gkg_edge.hip) against the reference's
literal form (gather to (B, 2C, N, k), grouped conv + BN + GELU, max) evaluated with torch ops on the same GPU.
    python tools/bench_edgeconv.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gkgnet_amd import layers
from gkgnet_amd.graph import EdgeConv2d

layers.norm_cfg["type"] = "BN"
for B, C, N, k in [(32, 64, 196, 9), (32, 320, 324, 9), (8, 80, 5184, 9)]:
    torch.manual_seed(0)
    mod = EdgeConv2d(C, 2 * C, "gelu", "batch", True).cuda().train()
    x = torch.randn(B, C, N, 1, device="cuda", requires_grad=True)
    idx = torch.randint(0, N, (B, N, k), device="cuda")
    edge = torch.stack([idx, torch.arange(N, device="cuda").view(1, N, 1).expand(B, N, k)])

    def run(hip):
        plan = EdgeConv2d._hip_plan
        if not hip:
            EdgeConv2d._hip_plan = lambda self, x: None
        try:
            for _ in range(3):
                mod(x, edge).sum().backward()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                mod(x, edge).sum().backward()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / 10 * 1e3
        finally:
            EdgeConv2d._hip_plan = plan
    torch.cuda.reset_peak_memory_stats(); th = run(True); mh = torch.cuda.max_memory_allocated() / 2**20
    torch.cuda.reset_peak_memory_stats(); tl = run(False); ml = torch.cuda.max_memory_allocated() / 2**20
    print(f"B={B} C={C} N={N} k={k}: HIP path {th:.2f} ms ({mh:.0f} MiB peak)   literal torch form {tl:.2f} ms ({ml:.0f} MiB peak)")
