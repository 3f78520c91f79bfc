#!/usr/bin/env python
"""One aggregation-backward shape under forced kernel flags, for counter collection and A/B timing:
    python tools/ubench/mr_bwd_shape.py s1 <flags-int> [iters]
(shape names and flag bits: tools/bench_mr_bwd.py / csrc/gkg_mr.hip gkg_mr_bwd_tm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gkgnet_amd import _lib
from gkgnet_amd.ops import _ptr, _stream
from tools.bench_mr_bwd import SHAPES


def main():
    name, flags = sys.argv[1], int(sys.argv[2], 0)
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    lib = _lib.load()
    torch.manual_seed(0)
    B, G, C, N, M, k = SHAPES[name]
    Mk = N if M is None else M
    base = (torch.arange(N, device="cuda") * Mk // N).view(1, N, 1)
    idx = ((base + torch.randint(-20, 21, (B * G, N, k), device="cuda")) % Mk).contiguous()
    sel = torch.randint(0, k, (B, N, C), device="cuda")
    arg = torch.gather(idx.view(B, G, N, k).permute(0, 2, 1, 3).reshape(B, N, G, 1, k).expand(B, N, G, C // G, k).reshape(B, N, C, k),
                       3, sel.unsqueeze(-1)).squeeze(-1).to(torch.int16).contiguous()
    g = torch.randn(B * N, 2 * C, device="cuda")
    gx = torch.empty(B, N, C, device="cuda")
    gs = None if M is None else torch.empty(B, Mk, C, device="cuda")
    if (flags >> 23) & 1:                      # measurement: the winning rows as 8-channel planes (B, C / 8, N, 8)
        arg = arg.view(B, N, C // 8, 8).permute(0, 2, 1, 3).contiguous()
    def call():
        _lib.check(lib.gkg_mr_bwd_tm(_ptr(g), _ptr(idx), _ptr(arg), _ptr(gx), _ptr(gs), B, G, C // G, N, Mk, k, 1, 1, flags, _stream()), "gkg_mr_bwd_tm")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record(); e1.synchronize()
    print(f"{name} flags={flags:#x}: {e0.elapsed_time(e1) / iters * 1000:.1f} us", flush=True)


if __name__ == "__main__":
    main()
