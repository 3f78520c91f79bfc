#!/usr/bin/env python
"""Aggregation backward (gkg_mr_bwd_tm) per shape: what the library picks ("rule"), the two-sweep exact 64-bit fixed-point scatter
("i64"), fp32 LDS atomics ("f32") and the one-sweep streaming forms (round 5: "s64" sampled-scale fixed point, "sf64" fp64 LDS
atomics) over forced chunk widths / workgroup sizes / rows in flight (flags bits 8..22, measurement only).
python tools/bench_mr_bwd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gkgnet_amd import _lib
from gkgnet_amd.ops import _ptr, _stream

# name: (B, G, C, N, M (None = self), k)
SHAPES = {"cfg2": (32, 4, 320, 324, None, 9), "cfg2_label": (32, 4, 320, 80, 324, 9), "cfg2ref": (32, 2, 640, 324, None, 9),
          "s3": (32, 2, 400, 1296, None, 9), "s2": (32, 2, 160, 5184, 1296, 9), "s1": (32, 2, 80, 20736, 1296, 9)}

def main():
    lib = _lib.load()
    torch.manual_seed(0)
    for name, (B, G, C, N, M, k) in SHAPES.items():
        Mk = N if M is None else M
        base = (torch.arange(N, device="cuda") * Mk // N).view(1, N, 1)
        idx = ((base + torch.randint(-20, 21, (B * G, N, k), device="cuda")) % Mk).contiguous()
        sel = torch.randint(0, k, (B, N, C), device="cuda")
        arg = torch.gather(idx.view(B, G, N, k).permute(0, 2, 1, 3).reshape(B, N, G, 1, k).expand(B, N, G, C // G, k).reshape(B, N, C, k),
                           3, sel.unsqueeze(-1)).squeeze(-1).to(torch.int16).contiguous()
        g = torch.randn(B * N, 2 * C, device="cuda")
        gx = torch.empty(B, N, C, device="cuda")
        gs = None if M is None else torch.empty(B, Mk, C, device="cuda")
        row = {}
        for kind, base_flags in (("rule", 0), ("i64", 3 << 16), ("f32", _lib.MR_FP32_ATOMICS)):
            for cw in ((0,) if kind == "rule" else (0, 4, 8, 16, 32, 64)):
                if cw and (C % cw or (C // G) % cw or Mk * cw * (8 if kind == "i64" else 4) + 16 > 96 * 1024):
                    continue
                flags = base_flags | (cw << 8)
                def call():
                    _lib.check(lib.gkg_mr_bwd_tm(_ptr(g), _ptr(idx), _ptr(arg), _ptr(gx), _ptr(gs), B, G, C // G, N, Mk, k, 1, 1, flags,
                                                 _stream()), "gkg_mr_bwd_tm")
                for _ in range(3):
                    call()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    call()
                e1.record(); e1.synchronize()
                row[kind if kind == "rule" else f"{kind}/{cw or 'rule'}"] = round(e0.elapsed_time(e1) * 100, 1)
        # streaming forms (round 5): flags bits 16..17 (1 sampled-scale fixed point, 2 fp64 atomics), 20..21 workgroup size
        ref_gx, ref_gs = gx.clone(), (gs.clone() if gs is not None else None)
        _lib.check(lib.gkg_mr_bwd_tm(_ptr(g), _ptr(idx), _ptr(arg), _ptr(ref_gx), _ptr(ref_gs), B, G, C // G, N, Mk, k, 1, 1,
                                     (3 << 16) if Mk <= 512 else _lib.MR_FP32_ATOMICS, _stream()), "ref")
        for sv, sname in ((1, "s64"), (2, "sf64")):
            for ntc, nt in (((1, 512), (2, 256)) if sv == 1 else ((1, 512),)):
                for cw, u8 in (((4, 0), (8, 0), (16, 0), (8, 1), (16, 1)) if sv == 1 else ((4, 0), (8, 0), (16, 0))):
                    if C % cw or (C // G) % cw or Mk * cw * 8 + 16 > 150 * 1024:
                        continue
                    flags = (sv << 16) | (ntc << 20) | (cw << 8) | (u8 << 22)
                    def call():
                        _lib.check(lib.gkg_mr_bwd_tm(_ptr(g), _ptr(idx), _ptr(arg), _ptr(gx), _ptr(gs), B, G, C // G, N, Mk, k, 1, 1, flags,
                                                     _stream()), "gkg_mr_bwd_tm")
                    gx.zero_()
                    if gs is not None:
                        gs.zero_()
                    for _ in range(3):
                        call()
                    torch.cuda.synchronize()
                    err = (gx - ref_gx).abs().max().item() if gs is None else max((gx - ref_gx).abs().max().item(), (gs - ref_gs).abs().max().item())
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        call()
                    e1.record(); e1.synchronize()
                    row[f"{sname}/{nt}/{cw}" + ("/u8" if u8 else "")] = (round(e0.elapsed_time(e1) * 100, 1), f"{err:.1e}")
        print(name, row, flush=True)

if __name__ == "__main__":
    main()
