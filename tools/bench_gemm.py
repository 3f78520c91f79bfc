#!/usr/bin/env python
"""Micro-benchmark + check of the fused projection kernels (csrc/gkg_gemm.hip) against torch.mm / torch.bmm at the
cfg2 shapes.   python tools/bench_gemm.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gkgnet_amd import _lib, fused
from gkgnet_amd.ops import _ptr, _stream

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / n

def main():
    lib = _lib.load()
    torch.manual_seed(0)
    dev = torch.device("cuda")
    shapes = [(10368, 320, 320, 1), (10368, 160, 160, 4), (10368, 640, 320, 1), (2560, 320, 320, 1), (2560, 160, 160, 4),
              (2560, 640, 320, 1), (2560, 320, 1280, 1), (2560, 1280, 320, 1), (41472, 400, 400, 1), (41472, 800, 400, 1)]
    for R, cin, cout, nb in shapes:
        x = torch.randn(nb, R, cin, device=dev)
        W = torch.randn(nb, cout, cin, device=dev) / cin ** 0.5
        bn = torch.nn.BatchNorm2d(nb * cout).to(dev).train()
        fl = 2.0 * R * cin * cout * nb
        Y, a, c, mean, invstd = fused._linear_fwd_own(lib, x, W, None, bn, R, cin, cout, nb)
        Yr = torch.bmm(x, W.transpose(1, 2))
        err = (Y.view(nb, R, cout) - Yr).abs().max().item()
        m_ref, v_ref = Yr.mean(1).flatten(), Yr.var(1, unbiased=False).flatten()
        merr = (mean - m_ref).abs().max().item()
        verr = ((1 / invstd ** 2 - 1e-5) - v_ref).abs().max().item() / v_ref.max().item()
        t_own = timeit(lambda: fused._linear_fwd_own(lib, x, W, None, bn, R, cin, cout, nb))
        t_lib = timeit(lambda: torch.bmm(x, W.transpose(1, 2)))
        # backward
        g = torch.randn(nb, R, cout, device=dev)
        dx, dW, dgam, dbet = fused._linear_bwd_own(lib, g, cout, R * cout, Y, a, c, mean, invstd, x, W, R, cin, cout, nb, 0, True)
        # reference: dy via the closed form, then library GEMMs
        yhat = (Yr - mean.view(nb, 1, cout)) * invstd.view(nb, 1, cout)
        s1 = g.sum(1, keepdim=True) / R; s2 = (g * yhat).sum(1, keepdim=True) / R
        dy = a.view(nb, 1, cout) * (g - s1 - yhat * s2)
        dx_r = torch.bmm(dy, W); dW_r = torch.bmm(dy.transpose(1, 2), x)
        ex = (dx.view(nb, R, cin) - dx_r).abs().max().item() / dx_r.abs().max().item()
        ew = (dW.view(nb, cout, cin) - dW_r).abs().max().item() / dW_r.abs().max().item()
        t_bwd = timeit(lambda: fused._linear_bwd_own(lib, g, cout, R * cout, Y, a, c, mean, invstd, x, W, R, cin, cout, nb, 0, True))
        t_bl = timeit(lambda: (torch.bmm(dy, W), torch.bmm(dy.transpose(1, 2), x)))
        print(f"R={R:6d} cin={cin:5d} cout={cout:5d} nb={nb}: fwd own {t_own:7.1f} us ({fl / t_own / 1e6:6.1f} TF) lib {t_lib:7.1f} us "
              f"({fl / t_lib / 1e6:6.1f} TF) | bwd own(stats+coef+dgrad+wgrad) {t_bwd:7.1f} us lib(2 GEMMs only) {t_bl:7.1f} us | "
              f"err y {err:.2e} mean {merr:.2e} var {verr:.2e} dx {ex:.2e} dW {ew:.2e}", flush=True)

if __name__ == "__main__":
    main()
