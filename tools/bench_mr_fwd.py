#!/usr/bin/env python
"""gkg_mr_fwd_tm per shape: HIP-event time of the kernel (library profiling scope) and fraction of 8 TB/s on the algorithmic bytes.
    python tools/bench_mr_fwd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gkgnet_amd import _lib

lib = _lib.load()
# name, B, G, c, N, M (None: self graph), mode, arg_kind
SHAPES = [("cfg2 grapher", 32, 4, 80, 324, None, 1, 1), ("cfg2 label", 32, 4, 80, 80, 324, 1, 1),
          ("stage1 (pooled 1296)", 32, 2, 40, 20736, 1296, 1, 1), ("stage2 (pooled 1296)", 32, 2, 80, 5184, 1296, 1, 1),
          ("stage3 self 1296", 32, 2, 200, 1296, None, 1, 1), ("stage4 self 324", 32, 2, 320, 324, None, 1, 1),
          ("stage1 label", 32, 2, 40, 80, 20736, 1, 1), ("mode 0 / slot arg", 8, 4, 48, 1000, 300, 0, 0)]


def run(x, src, idx, B, G, c, N, M, mode, ak, k=9):
    C = G * c
    out = torch.empty((B * N, 2 * C) if (mode & 0xff) == 1 else (B, N, C), device="cuda")
    arg = torch.empty((B, N, C), dtype=torch.int16 if ak else torch.uint8, device="cuda")

    def call():
        _lib.check(lib.gkg_mr_fwd_tm(x.data_ptr(), 0, 0, None if src is None else src.data_ptr(), idx.data_ptr(), out.data_ptr(), arg.data_ptr(),
                                     B, G, c, N, M, k, mode, 0, ak, None), "gkg_mr_fwd_tm")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(20):
        call()
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    ms, cnt = _lib.prof_read()["mr_fwd"]
    work = _lib.prof_work("mr_fwd") / max(cnt, 1)
    return ms / cnt * 1e3, work, out, arg


def main():
    torch.manual_seed(0)
    for name, B, G, c, N, M, mode, ak in SHAPES:
        C = G * c
        Mk = N if M is None else M
        x = torch.randn(B, N, C, device="cuda")
        src = None if M is None else torch.randn(B, Mk, C, device="cuda")
        base = (torch.arange(N, device="cuda") * Mk // N).view(1, N, 1)
        idx = ((base + torch.randint(-40, 41, (B * G, N, 9), device="cuda")) % Mk).contiguous()
        t0, work, o0, a0 = run(x, src, idx, B, G, c, N, Mk, mode, ak)
        line = f"{name:22s} B={B} G={G} c={c} N={N} M={Mk}: {t0:7.1f} us ({work / t0 / 8e6:.2f} of HBM)"
        xn = x.clone(); xn[0, 3, 5] = float("nan"); xn[1, 7, 2] = float("inf")     # non-finite inputs: the careful chain, same time class
        t1, _, _, _ = run(xn, src, idx, B, G, c, N, Mk, mode, ak)
        line += f" | with a NaN and an inf in x: {t1:7.1f} us"
        print(line, flush=True)


if __name__ == "__main__":
    main()
