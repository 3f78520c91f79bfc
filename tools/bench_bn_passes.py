#!/usr/bin/env python
"""The bandwidth passes between the projections (train-mode BN statistics / apply + GELU / backward statistics / backward
apply) at GKGNet-576 stage shapes, each launch timed with HIP events through torch.profiler, against the bytes it must move.
    python tools/bench_bn_passes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from gkgnet_amd import fused

SHAPES = [(663552, 320), (663552, 80), (165888, 640), (41472, 1600), (10368, 1280)]
for T, C in SHAPES:
    y = torch.randn(T, C, device="cuda", requires_grad=True)
    bn = torch.nn.BatchNorm2d(C).cuda().train()
    act = torch.nn.GELU()
    g = torch.randn(T, C, device="cuda")
    for _ in range(2):
        out = fused._BNActTM.apply(y, bn.weight, bn.bias, bn, 1)
        out.backward(g)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(3):
            out = fused._BNActTM.apply(y, bn.weight, bn.bias, bn, 1)
            out.backward(g)
        torch.cuda.synchronize()
    mb = T * C * 4 / 1e6
    rows = {}
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CPU and "gkg::" in ev.name:
            rows.setdefault(ev.name.split("(")[0][:48], []).append(ev.device_time)
    print(f"T={T} C={C} ({mb:.0f} MB per tensor)")
    need = {"col_stats": 1, "bn_stats": 1, "affine_act": 2, "bn_bwd_stats": 2, "bn_bwd_apply": 3, "reduce": 0}
    for n, v in rows.items():
        us = sum(v) / len(v)
        k = next((m for kk, m in need.items() if kk in n), 0)
        print(f"   {n:50s} {us:8.1f} us" + (f"   {k} x tensor = {k * mb:.0f} MB -> {k * mb / us:.2f} TB/s" if k else ""))
