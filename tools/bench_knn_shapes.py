#!/usr/bin/env python
"""HIP-event time of the k-NN tile kernel alone (the library's profiling scopes) at short-stream shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gkgnet_amd import ops, _lib

SHAPES = [("pvig_s stage1 +rp", 64, 40, 20736, 1296, "xy"), ("pvig_s stage1 no rp", 64, 40, 20736, 1296, False),
          ("pvig_s stage2 +rp", 64, 80, 5184, 1296, "xy"), ("pvig_s stage2 no rp", 64, 80, 5184, 1296, False),
          ("cfg2 grapher", 128, 80, 324, 324, True), ("cfg2 label", 128, 80, 80, 324, False), ("s4 grapher c=320", 64, 320, 324, 324, True),
          ("grapher 24x24", 128, 80, 576, 576, True), ("label over 1296", 128, 80, 80, 1296, False), ("c=40 18x18", 256, 40, 324, 324, True)]


def main():
    torch.manual_seed(0)
    for name, BG, c, N, M, rp in SHAPES:
        x = torch.randn(BG, c, N, device="cuda")
        y = None if rp is True else torch.randn(BG, c, M, device="cuda")
        r = -torch.rand(1, N, M, device="cuda") if rp else None
        f = lambda: ops.knn_graph(x, y, r, 9, 1)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        _lib.prof_reset(); _lib.prof_enable(True)
        for _ in range(30):
            f()
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        pr = _lib.prof_read()
        print(f"   {name:20s} BG={BG} c={c} N={N} M={M}: knn_tile scope {pr['knn_tile'][0] / max(pr['knn_tile'][1], 1) * 1e3:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
