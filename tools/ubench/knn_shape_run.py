#!/usr/bin/env python
"""Run the library's k-NN at one of pvig_m's stage shapes a few times (a target for rocprofv3 --pmc / --kernel-trace):
python tools/ubench/knn_shape_run.py s1|s2|s3 [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gkgnet_amd import ops

SH = {"s1": (128, 12, 36864, 2304, 18, 1, True), "s2": (128, 24, 9216, 2304, 18, 1, True), "s3": (128, 48, 2304, 2304, 18, 2, False)}
BG, c, N, M, k, dil, pooled = SH[sys.argv[1]]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.manual_seed(0)
x = torch.randn(BG, c, N, device="cuda")
y = torch.randn(BG, c, M, device="cuda") if pooled else None
r = -torch.rand(1, N, M, device="cuda")
from gkgnet_amd import _lib
for _ in range(2):
    ops.knn_graph(x, y, r, k, dil)
torch.cuda.synchronize()
_lib.prof_reset(); _lib.prof_enable(True)
for _ in range(reps):
    ops.knn_graph(x, y, r, k, dil)
torch.cuda.synchronize()
_lib.prof_enable(False)
pr = _lib.prof_read()
print(f"{sys.argv[1]}: BG={BG} c={c} N={N} M={M} k={k} d={dil}: knn_tile scope {pr['knn_tile'][0] / max(pr['knn_tile'][1], 1) * 1e3:8.1f} us", flush=True)
