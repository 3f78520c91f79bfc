#!/usr/bin/env python
"""Which aten / autograd operators of a GKGNet-576 training step (cfg4: forward + backward of the backbone, fp32) or inference
forward (cfg3: eval, bf16 autocast) launch the torch-side
glue kernels (strided elementwise, adds, reductions, copies): torch.profiler, one eager step; per kernel-name pattern the CPU
operators above the launches, with counts and total time.     python tools/prof_backbone_ops.py [cfg3|cfg4] [pattern ...]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gkgnet_amd.backbone import GKGNet
from torch.profiler import profile, ProfilerActivity

pats = sys.argv[1:] or ["elementwise_kernel_manual_unroll", "CUDAFunctor_add", "reduce_kernel", "Memcpy", "vectorized_elementwise"]
MODE = "cfg3" if "cfg3" in sys.argv else ("cfg5" if "cfg5" in sys.argv else "cfg4")
pats = [a for a in pats if a not in ("cfg3", "cfg4", "cfg5")] or ["elementwise_kernel_manual_unroll", "CUDAFunctor_add", "reduce_kernel", "Memcpy", "vectorized_elementwise", "copy_kernel", "transpose", "MIOpen"]
spec = bench.BACKBONE_WORKLOADS[MODE]
torch.manual_seed(0)
net = GKGNet(**dict(spec["kw"])).cuda()
net = net.eval() if MODE != "cfg4" else net.train()
img = torch.randn(spec["B"], 3, spec["kw"]["size"], spec["kw"]["size"], device="cuda")
if MODE == "cfg4":
    img = img.contiguous(memory_format=torch.channels_last)


def step():
    if MODE != "cfg4":                       # the bench's forward: eval, bf16 autocast
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            net(img)
        return
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
