#!/usr/bin/env python
"""How torch runs  out_f32 = x_bf16 @ W_bf16^T + bias  (the Grapher fc1 of the bf16 inference path): which kernels each
spelling launches and what they cost at the GKGNet-576 stage shapes.    python tools/ubench/addmm_f32out.py"""
import torch
from torch.profiler import profile, ProfilerActivity

F32, BF = torch.float32, torch.bfloat16
for R, K, N in ((663552, 80, 80), (165888, 160, 160), (41472, 400, 400), (10368, 640, 640)):
    x = torch.randn(R, K, device="cuda", dtype=BF)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(BF)
    b32 = torch.randn(N, device="cuda")
    b16 = b32.to(BF)
    out = torch.empty(R, N, device="cuda", dtype=F32)
    variants = {
        "addmm(bias f32, out_dtype=f32)        [shipped]": lambda: torch.addmm(b32, x, w.t(), out_dtype=F32),
        "addmm(bias bf16, out_dtype=f32)": lambda: torch.addmm(b16, x, w.t(), out_dtype=F32),
        "mm(out_dtype=f32)                     [no bias]": lambda: torch.mm(x, w.t(), out_dtype=F32),
        "mm(out_dtype=f32) ; add_(bias)": lambda: torch.mm(x, w.t(), out_dtype=F32).add_(b32),
        "_addmm_activation(bias bf16) -> bf16  [for scale]": lambda: torch._addmm_activation(b16, x, w.t(), use_gelu=False),
    }
    print(f"== {R} x {K} -> {N}")
    for name, fn in variants.items():
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); e1.synchronize()
            with profile(activities=[ProfilerActivity.CUDA]) as prof:
                fn()
                torch.cuda.synchronize()
            ks = [(e.key[:48], round(e.device_time_total, 1)) for e in prof.key_averages() if e.device_time_total > 0]
            print(f"  {name:52s} {e0.elapsed_time(e1) * 50:7.1f} us   {ks}")
        except Exception as exc:
            print(f"  {name:52s} failed: {type(exc).__name__}: {str(exc)[:100]}")
