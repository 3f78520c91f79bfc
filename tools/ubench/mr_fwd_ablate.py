#!/usr/bin/env python
"""Compile-time ablation of mr_fwd_tm_kernel (csrc/gkg_mr.hip, -DMR_ABL=<bits>): which of index loads / row gathers / stores
bounds the gather at the stage shapes.  python tools/ubench/mr_fwd_ablate.py"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

CS = os.path.join(ROOT, "gkgnet_amd", "csrc")
VARIANTS = {"full": 0, "fixed_rows": 1, "no_idx_loads": 2, "no_idx_fixed_rows": 3, "no_out": 4, "no_arg": 8, "no_stores": 12, "loads_only": 12,
            "skeleton": 15}
SHAPES = [("cfg2 grapher", 32, 4, 80, 324, None), ("stage1", 32, 2, 40, 20736, 1296), ("stage2", 32, 2, 80, 5184, 1296), ("stage3", 32, 2, 200, 1296, None)]


def build(bits, *extra):
    so = f"/tmp/libmr_{bits}{'_'.join(e.strip('-').replace('=', '') for e in extra)}.so"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           f"-DMR_ABL={bits}", *extra, "-I" + os.path.join(ROOT, "include"), "-I" + CS, "-o", so,
                           os.path.join(CS, "gkg_mr.hip"), os.path.join(CS, "gkg_api.hip")])
    lib = C.CDLL(so)
    lib.gkg_mr_fwd_tm.restype = C.c_int
    lib.gkg_mr_fwd_tm.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 9 + [C.c_void_p]
    return lib


def main():
    torch.manual_seed(0)
    data = []
    for name, B, G, c, N, M in SHAPES:
        Cc = G * c
        Mk = N if M is None else M
        x = torch.randn(B, N, Cc, device="cuda")
        src = None if M is None else torch.randn(B, Mk, Cc, device="cuda")
        base = (torch.arange(N, device="cuda") * Mk // N).view(1, N, 1)
        idx = ((base + torch.randint(-40, 41, (B * G, N, 9), device="cuda")) % Mk).contiguous()
        out = torch.empty(B * N, 2 * Cc, device="cuda")
        arg = torch.zeros(B * N * Cc + 512 * 1024, dtype=torch.int16, device="cuda")       # + room for the timeline stamps
        data.append((name, x, src, idx, out, arg, B, G, c, N, Mk))
    for v, bits in list(VARIANTS.items()):
        if len(sys.argv) > 1 and v not in sys.argv[1:]:
            continue
        lib = build(*(bits.split() if isinstance(bits, str) else [bits]))
        line = f"{v:18s}"
        for name, x, src, idx, out, arg, B, G, c, N, Mk in data:
            def call():
                rc = lib.gkg_mr_fwd_tm(x.data_ptr(), 0, 0, None if src is None else src.data_ptr(), idx.data_ptr(), out.data_ptr(), arg.data_ptr(),
                                       B, G, c, N, Mk, 9, 1, 0, 1, None)
                assert rc == 0
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                call()
            e1.record(); e1.synchronize()
            line += f"  {name} {e0.elapsed_time(e1) * 50:7.1f}"
        print(line, flush=True)


def timeline():
    """-DMR_TL: per sampled wave, cycles from its start to: index + centre loads back, row gathers back, stores issued, stores
    acknowledged"""
    import numpy as np
    lib = build(0, "-DMR_TL")
    torch.manual_seed(0)
    for name, B, G, c, N, M in SHAPES:
        Cc = G * c
        Mk = N if M is None else M
        x = torch.randn(B, N, Cc, device="cuda")
        src = None if M is None else torch.randn(B, Mk, Cc, device="cuda")
        base = (torch.arange(N, device="cuda") * Mk // N).view(1, N, 1)
        idx = ((base + torch.randint(-40, 41, (B * G, N, 9), device="cuda")) % Mk).contiguous()
        out = torch.empty(B * N, 2 * Cc, device="cuda")
        arg = torch.zeros(B * N * Cc + 512 * 1024, dtype=torch.int16, device="cuda")
        for _ in range(3):
            arg[B * N * Cc:].zero_()
            rc = lib.gkg_mr_fwd_tm(x.data_ptr(), 0, 0, None if src is None else src.data_ptr(), idx.data_ptr(), out.data_ptr(), arg.data_ptr(),
                                   B, G, c, N, Mk, 9, 1, 0, 1, None)
            assert rc == 0
            torch.cuda.synchronize()
        st = arg[B * N * Cc:].view(torch.int64).cpu().numpy().reshape(-1, 8)
        st = st[st[:, 0] > 0]
        d = st[:, 1:5] - st[:, :1]
        print(f"{name}: {len(st)} sampled waves; median cycles from wave start to [idx+x back, gathers back, stores issued, stores acked] = "
              f"{np.median(d, axis=0).astype(int).tolist()}, p90 = {np.percentile(d, 90, axis=0).astype(int).tolist()}", flush=True)


if __name__ == "__main__":
    if "timeline" in sys.argv:
        timeline()
    else:
        main()
