#!/usr/bin/env python
"""Which aten / autograd operators of a GKGNet-576 training step (forward + backward of the backbone, fp32) launch the torch-side
glue kernels (strided elementwise, adds, reductions, copies): torch.profiler, one eager step; per kernel-name pattern the CPU
operators above the launches, with counts and total time.     python tools/prof_backbone_ops.py [pattern ...]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gkgnet_amd.backbone import GKGNet
from torch.profiler import profile, ProfilerActivity

pats = sys.argv[1:] or ["elementwise_kernel_manual_unroll", "CUDAFunctor_add", "reduce_kernel", "Memcpy", "vectorized_elementwise"]
spec = bench.BACKBONE_WORKLOADS["cfg4"]
torch.manual_seed(0)
net = GKGNet(**dict(spec["kw"])).cuda().train()
img = torch.randn(spec["B"], 3, 576, 576, device="cuda").contiguous(memory_format=torch.channels_last)


def step():
    out = net(img)
    outs = out if isinstance(out, (tuple, list)) else (out,)
    sum(o.float().sum() for o in outs if torch.is_tensor(o)).backward()
    net.zero_grad(set_to_none=True)


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
evs = prof.events()
cpu = [ev for ev in evs if ev.device_type == torch.autograd.DeviceType.CPU]
agg = collections.defaultdict(lambda: [0, 0.0])
for c in cpu:
    for k in c.kernels:
        if any(p in k.name for p in pats):
            key = (next(p for p in pats if p in k.name), c.name, str(list(c.input_shapes))[:110])
            agg[key][0] += 1
            agg[key][1] += k.duration
for (pat, op, shapes), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{us:9.1f} us x{n:4d}  {pat[:34]:34s} <- {op} {shapes}")
