#!/usr/bin/env python
"""Compile-time phase ablation of the fused aggregation + projection TRAINING kernel (csrc/gkg_mrgemm_x6.hip): builds private
copies with -DMX_ABL=<bits> (and optional extra -D flags) into /tmp and times gkg_mr_linear_x6 with HIP events.
    python tools/ubench/mrgemm_x6_ablate.py [variant ...]"""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

CS = os.path.join(ROOT, "gkgnet_amd", "csrc")
SHAPES = {"cfg2": (32, 4, 320, 324, None, 9), "cfg2_label": (32, 4, 320, 80, 324, 9), "s3": (32, 2, 400, 1296, None, 9),
          "s4": (32, 2, 640, 324, None, 9), "s1": (32, 2, 80, 20736, 1296, 9)}
VARIANTS = {"full": 0, "no_gather": 1, "no_mfma": 2, "no_wload": 4, "no_idx": 8, "no_store": 16, "no_stats": 32, "no_split": 64,
            "no_gather_idx": 9, "mfma_only": 1 + 8 + 16 + 32 + 64, "skeleton": 127}


def build(bits, extra=()):
    tag = f"{bits}_" + "_".join(e.replace("=", "") for e in extra)
    so = f"/tmp/libmx_{tag}.so"
    srcs = [os.path.join(CS, f) for f in ("gkg_mrgemm_x6.hip", "gkg_api.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           f"-DMX_ABL={bits}", "-Wno-pass-failed", "-I" + os.path.join(ROOT, "include"), "-I" + CS, "-o", so] +
                          ["-D" + e for e in extra] + srcs)
    lib = C.CDLL(so)
    lib.gkg_mr_linear_x6.restype = C.c_int
    lib.gkg_mr_linear_x6.argtypes = [C.c_void_p] * 8 + [C.c_int] * 6 + [C.c_void_p]
    return lib


def main():
    torch.manual_seed(0)
    want = sys.argv[1:] or list(VARIANTS)
    data = {}
    for name, (B, G, Cc, N, M, k) in SHAPES.items():
        x = torch.randn(B, N, Cc, device="cuda")
        src = None if M is None else torch.randn(B, M, Cc, device="cuda")
        Mk = N if M is None else M
        base = (torch.arange(N, device="cuda") * Mk // N).view(1, N, 1)
        idx = ((base + torch.randint(-20, 21, (B * G, N, k), device="cuda")) % Mk).contiguous()
        ci = Cc // 2
        planes = torch.zeros(4 * 3 * ((ci + 31) // 32 * 4) * ((ci + 127) // 128 * 128) * 16, dtype=torch.uint8, device="cuda")   # [4][3][KC][NP] x 16 B
        planes.view(torch.bfloat16).normal_()
        T = B * N
        y = torch.empty(4, T, ci, device="cuda")
        u = torch.empty(4, T, ci, device="cuda")
        arg = torch.empty(B, N, Cc, dtype=torch.int16, device="cuda")
        sums = torch.zeros(4 * 2 * ci, dtype=torch.float64, device="cuda")
        data[name] = (x, src, idx, planes, y, arg, u, sums, B, G, Cc // G, N, Mk, k)
    res = {n: {} for n in SHAPES}
    for v in want:
        extra = ()
        if "+" in v:
            v, *extra = v.split("+")
        lib = build(VARIANTS[v], tuple(extra))
        for name, (x, src, idx, planes, y, arg, u, sums, B, G, cg, N, Mk, k) in data.items():
            for save_u in (True, False):
                def call():
                    rc = lib.gkg_mr_linear_x6(x.data_ptr(), None if src is None else src.data_ptr(), idx.data_ptr(), planes.data_ptr(),
                                              y.data_ptr(), arg.data_ptr(), u.data_ptr() if save_u else None, sums.data_ptr(),
                                              B, G, cg, N, Mk, k, None)
                    assert rc == 0, rc
                for _ in range(5):
                    call()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    call()
                e1.record(); e1.synchronize()
                res[name][v + ("+" + "+".join(extra) if extra else "") + ("" if save_u else "/nou")] = round(e0.elapsed_time(e1) * 50, 1)
    for n, r in res.items():
        print(n, r, flush=True)


if __name__ == "__main__":
    main()
