#!/usr/bin/env python
"""Compile-time ablations of knn_pf_kernel (-DPF_ABL bits, csrc/gkg_knn_pf.hip) at pvig_s stage shapes: what the prefilter
launch pays for.  Builds one k-NN-only library per variant in /tmp on the GPU box.   python tools/ubench/pf_ablate.py [bits ...]"""
import ctypes as C, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

CS = os.path.join(ROOT, "gkgnet_amd", "csrc")
SRCS = ["gkg_api.hip", "gkg_knn.hip", "gkg_knn_f32.hip", "gkg_knn_f32_norp.hip", "gkg_knn_f32_mr.hip", "gkg_knn_f32_mr_norp.hip",
        "gkg_knn_bf.hip", "gkg_knn_bf_norp.hip", "gkg_knn_pf.hip", "gkg_knn_pf_norp.hip"]
NAMES = {0: "shipped", 1: "no appends (tested, never stored)", 2: "no MFMAs", 4: "no relative_pos loads", 8: "no selection",
         16: "no key loads", 64: "flush counters"}


def build(abl):
    so = f"/tmp/libgkg_pf_abl{abl}.so"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
             "-Wno-pass-failed", f"-DPF_ABL={abl}", "-DKNN_ABLATE=99", "-I" + os.path.join(ROOT, "include"), "-I" + CS]
    objs = [f"/tmp/pfa{abl}_{s}.o" for s in SRCS]
    with ThreadPoolExecutor(10) as ex:
        list(ex.map(lambda so_: subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", os.path.join(CS, so_[0]), "-o", so_[1]]),
                    zip(SRCS, objs)))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    lib = C.CDLL(so)
    lib.gkg_knn_workspace_bytes.restype = C.c_size_t
    lib.gkg_knn_workspace_bytes.argtypes = [C.c_int] * 7 + [C.c_uint]
    lib.gkg_knn_fwd.restype = C.c_int
    lib.gkg_knn_fwd.argtypes = [C.c_void_p] * 5 + [C.c_int] * 7 + [C.c_uint, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.gkg_debug_set_knn_timeline.argtypes = [C.c_void_p]
    return lib


def main():
    variants = [int(a, 0) for a in sys.argv[1:]] or [0, 1, 2, 4, 8, 16, 1 | 2, 2 | 4 | 16]
    torch.manual_seed(0)
    shapes = (("pvig_s stage 1", 64, 40, 20736, 1296, 1), ("pvig_s stage 2", 64, 80, 5184, 1296, 1),
              ("pvig_s stage 3 (self graph, d = 2)", 64, 200, 1296, 1296, 2))
    data = {}
    for name, BG, c, N, M, dil in shapes:
        data[name] = (torch.randn(BG, c, N, device="cuda"), None if dil > 1 else torch.randn(BG, c, M, device="cuda"),
                      -torch.rand(N, M, device="cuda"), torch.empty(BG, N, 9, dtype=torch.int64, device="cuda"))
    for abl in variants:
        lib = build(abl)
        for name, BG, c, N, M, dil in shapes:
            x, y, r, idx = data[name]
            flags = 1 | 64                                  # GKG_KNN_NORMALIZE | GKG_KNN_RELPOS_UNIT
            nb = lib.gkg_knn_workspace_bytes(BG, c, N, M, 9, dil, 0, flags)
            ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
            ctr = torch.zeros(64, dtype=torch.int64, device="cuda")
            lib.gkg_debug_set_knn_timeline(ctr.data_ptr())
            def call():
                assert lib.gkg_knn_fwd(x.data_ptr(), None if y is None else y.data_ptr(), r.data_ptr(), idx.data_ptr(), None, BG, c, N, M, 9, dil, 0, flags,
                                       ws.data_ptr(), nb, None) == 0
            for _ in range(2):
                call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                call()
            e1.record(); e1.synchronize()
            label = " + ".join(NAMES[b] for b in NAMES if b and (abl & b)) or NAMES[0]
            print(f"{name}: PF_ABL={abl:2d} ({label}): {e0.elapsed_time(e1) / 5 * 1000:.0f} us (preparation + prefilter + clean-up)", flush=True)
            if abl & 64:
                f, e, r = [int(v) / 7 for v in ctr[:3].tolist()]          # 7 calls
                waves = BG * ((N + 63) // 64) * 4
                print(f"   per wave: {f / waves:.1f} flushes, {e / waves / 64:.1f} entries per lane, {r / waves:.1f} insert rounds "
                      f"({(M + 31) // 32 / 4:.1f} key tiles per wave, {M / 4:.0f} candidates per lane)", flush=True)


if __name__ == "__main__":
    main()
