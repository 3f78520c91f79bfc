#!/usr/bin/env python
"""Which aten / autograd operators of the eager cfg2 step launch the small glue kernels (copies, fills, elementwise adds):
torch.profiler with stacks over 3 eager steps; prints, per GPU kernel name matching a pattern, the CPU operators above it.
    python tools/prof_cfg2_ops.py [pattern ...]      (default: copyBuffer Fill add)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gkgnet_amd import parallel
from torch.profiler import profile, ProfilerActivity

pats = sys.argv[1:] or ["copyBuffer", "Memcpy", "Memset", "Fill", "CUDAFunctor_add", "reduce_kernel"]
w = bench.WORKLOADS["cfg2"]
dev = torch.device("cuda")
torch.manual_seed(0)
grapher, label = bench.build_modules(w, dev)
params = list(grapher.parameters()) + list(label.parameters())
bucket = parallel.GradBucket(params)
B, C, H, L = 32, w["C"], w["H"], w["L"]
x = torch.randn(B, C, H, H, device=dev).requires_grad_(True)
e = torch.randn(B, L, C, device=dev).requires_grad_(True)
cot_x = torch.randn(B, C, H, H, device=dev); cot_e = torch.randn(B, L, C, device=dev)


def compute():
    bucket.release(prezero=True)
    x.grad = None; e.grad = None
    out = grapher(x)
    e2, _ = label(e, out)
    torch.autograd.backward([out, e2], [cot_x, cot_e])
    bucket.pack()


for _ in range(3):
    compute()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    compute()
    torch.cuda.synchronize()
evs = prof.events()
cpu = [ev for ev in evs if ev.device_type == torch.autograd.DeviceType.CPU]
for ev in evs:
    if ev.device_type != torch.autograd.DeviceType.CPU and any(p in ev.name for p in pats):
        # the CPU ops whose time range contains the launch (correlated by time of the runtime launch call)
        t = ev.time_range.start
        par = [c for c in cpu if c.kernels and any(k.name == ev.name and abs(k.duration - ev.device_time) < 1e-3 for k in c.kernels)]
        if not par:      # memcpy / memset: the innermost CPU operators running when it was enqueued
            par = sorted([c for c in cpu if c.time_range.start <= t <= c.time_range.end], key=lambda c: c.time_range.end - c.time_range.start)[:3]
        names = sorted(set(f"{c.name}{list(c.input_shapes) if c.input_shapes else ''}" for c in par), key=len)[:4]
        print(f"{ev.name[:60]:60s} {ev.device_time:7.1f} us  <- {names}")
